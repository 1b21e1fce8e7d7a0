// mc_tables.h -- the triangle table of marching cubes (Lorensen & Cline 1987) in the widely used
// public-domain tabulation of P. Bourke / C. G. Bloyd ("Polygonising a scalar field", 1994), which
// is the one the reference indexes (src/core/cuda/TSDF.cu:67, triTable[cubeClass][i]).
//
// Conventions (the reference's, TSDF.cu:893-902 and 989-1096):
//   corner c of a cube at (x, y, z):  0:(0,0,0) 1:(1,0,0) 2:(1,0,1) 3:(0,0,1)
//                                     4:(0,1,0) 5:(1,1,0) 6:(1,1,1) 7:(0,1,1)      (dx, dy, dz)
//   cube class bit c = tsdf(corner c) < 0
//   edge e joins emf_mc_edge_corner[e][0] -> emf_mc_edge_corner[e][1]; an edge carries a vertex iff its corners'
//   class bits differ (that is all edgeTable[] encodes, so it is computed, not tabulated)
//   row `cls` lists edges three at a time, one triangle each.  Rows are written as strings of
//   edge digits (a = 10, b = 11) and expanded to the usual 256 x 16 layout, -1 terminated.
// tests/test_mc_tables.py checks every row (only active edges, every active edge used, closed
// fans, watertight unions over random fields) and the digest of the whole table against the
// fixture generated from the reference.
#pragma once

#define EMF_MC_ROWS                                                                                  \
    "", "083", "019", "183981", "12a", "08312a", "92a029", "2832a8a98", "3b2", "0b28b0", "19023b",   \
    "1b219b98b", "3a1ba3", "0a108a8ba", "3903b9ba9", "98aa8b", "478", "430734", "019847",            \
    "419471731", "12a847", "34730412a", "92a902847", "2a9297273794", "8473b2", "b47b24204",          \
    "90184723b", "47b94b9b2921", "3a13ba784", "1ba14b1047b4", "47890b9bab03", "47b4b99ba", "954",    \
    "954083", "054150", "854835315", "12a954", "30812a495", "52a542402", "2a5325354348", "95423b",   \
    "0b208b495", "05401523b", "21525828b485", "a3ba13954", "4950818a18ba", "54050b5bab03",           \
    "54858aa8b", "978579", "930953573", "078017157", "153357", "978957a12", "a12950530573",          \
    "802825857a52", "2a5253357", "7957893b2", "95797292027b", "23b018178157", "b21b17715",           \
    "958857a13a3b", "5705097b010aba0", "ba0b03a50807570", "ba57b5", "a65", "0835a6", "9015a6",       \
    "1831985a6", "165261", "165126308", "965906026", "598582526328", "23ba65", "b08b20a65",          \
    "01923b5a6", "5a61929b298b", "63b653513", "08b0b50515b6", "3b6036065059", "65969bb98",           \
    "5a6478", "43047365a", "1905a6847", "a65197173794", "612651478", "125526304347",                 \
    "847905065026", "739794329596269", "3b2784a65", "5a647242027b", "01947823b5a6",                  \
    "9219b294b7b45a6", "8473b53515b6", "51b5b610b7b404b", "059065036b63847", "65969b4797b9",         \
    "a4964a", "4a649a083", "a01a60640", "83181686461a", "149124264", "308129249264", "024426",       \
    "832824426", "a49a64b23", "08228b49a4a6", "3b201606461a", "64161a48121b8b1", "964936913b63",     \
    "8b1810b61914641", "3b6360064", "648b68", "7a678a89a", "0730a709a67a", "a671a7178180",           \
    "a67a71173", "126168189867", "269291679093739", "780706602", "732672", "23ba68a89867",           \
    "20727b09767a9a7", "1801781a767a23b", "b21b17a61671", "896867916b63136", "091b67",               \
    "7807063b0b60", "7b6", "76b", "308b76", "019b76", "819831b76", "a126b7", "12a3086b7",            \
    "2902a96b7", "6b72a3a83a98", "723627", "708760620", "276237019", "162186198876", "a76a17137",    \
    "a7617a187108", "03707a0a96a7", "76a7a88a9", "684b86", "36b306046", "86b846901",                 \
    "946963931b36", "6846b82a1", "12a30b06b046", "4b846b0292a9", "a93a32943b36463", "823842462",     \
    "042462", "190234246438", "194142246", "8138618466a1", "a10a06604", "4634386a3039a93",           \
    "a946a4", "49576b", "083495b76", "50154076b", "b76834354315", "954a1276b", "6b712a083495",       \
    "76b54a42a402", "348354325a52b76", "723762549", "954086062687", "362376150540",                  \
    "628687218485158", "954a16176137", "16a176107870954", "40a4a503a6a737a", "76a7a854a48a",         \
    "6956b9b89", "36b063056095", "0b805b01556b", "6b3635531", "12a95b9b8b56", "0b306b09656912a",     \
    "b85b56805a52025", "6b36352a3a53", "589528562382", "956960062", "158180568382628", "156216",     \
    "13616a386569896", "a10a06950560", "03856a", "a56", "b5a75b", "b5ab75830", "5b75ab190",          \
    "a75ab7981831", "b12b71751", "08312717572b", "9759279022b7", "75272b592328982", "25a235375",     \
    "820852875a25", "9015a35373a2", "982921872a25752", "135375", "087071175", "903935537",           \
    "987597", "5845a8ab8", "5045b05abb30", "01984a8aba45", "ab4a45b34941314", "2512852b8458",        \
    "04b0b345b2b151b", "0250592b5458b85", "9452b3", "25a352345384", "5a2524420", "3a235a385458019",  \
    "5a2524192942", "845853351", "045105", "845853905035", "945", "4b749b9ab", "0834979b79ab",       \
    "1ab1b414074b", "3143481a474bab4", "4b79b492b912", "9749b791b2b1083", "b74b42240",               \
    "b74b42834324", "29a279237749", "9a7974a27870207", "37a3a274a1a040a", "1a2874", "491417713",     \
    "491417081871", "403743", "487", "9a8ab8", "30939bb9a", "01a0a88ab", "31ab3a", "12b1b99b8",      \
    "30939b1292b9", "02b80b", "32b", "23828aa89", "9a2092", "23828a0181a8", "1a2", "138918", "091",  \
    "038", "",                                                                                       \

#ifdef __cplusplus
#define EMF_MC_CONST constexpr
#else
#define EMF_MC_CONST static const
#endif
EMF_MC_CONST signed char emf_mc_edge_corner_init[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6},
                                                            {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

#ifdef __cplusplus
// C++ (device code): expanded at compile time into constant memory.
struct EmfMcTriTable {
    signed char v[256][16];
};
struct EmfMcEdgeTable {
    signed char v[12][2];
};
constexpr EmfMcTriTable emf_mc_expand() {
    constexpr const char* rows[256] = {EMF_MC_ROWS};
    EmfMcTriTable t{};
    for (int c = 0; c < 256; ++c) {
        int i = 0;
        for (const char* p = rows[c]; *p; ++p) t.v[c][i++] = static_cast<signed char>(*p <= '9' ? *p - '0' : *p - 'a' + 10);
        for (; i < 16; ++i) t.v[c][i] = -1;
    }
    return t;
}
constexpr EmfMcEdgeTable emf_mc_edges() {
    EmfMcEdgeTable t{};
    for (int e = 0; e < 12; ++e) {
        t.v[e][0] = emf_mc_edge_corner_init[e][0];
        t.v[e][1] = emf_mc_edge_corner_init[e][1];
    }
    return t;
}
#ifdef __HIPCC__
__device__ __constant__ const EmfMcTriTable emf_mc_tri = emf_mc_expand();
__device__ __constant__ const EmfMcEdgeTable emf_mc_edge = emf_mc_edges();
#define emf_mc_tri_table emf_mc_tri.v
#define emf_mc_edge_corner emf_mc_edge.v
#endif
#else
// C (the CPU oracle): expanded on first use.
static const char* const emf_mc_rows[256] = {EMF_MC_ROWS};
static signed char emf_mc_tri_table[256][16];
#define emf_mc_edge_corner emf_mc_edge_corner_init
static void emf_mc_init(void) {
    static int done = 0;
    if (done) return;
    for (int c = 0; c < 256; ++c) {
        int i = 0;
        for (const char* p = emf_mc_rows[c]; *p; ++p)
            emf_mc_tri_table[c][i++] = (signed char)(*p <= '9' ? *p - '0' : *p - 'a' + 10);
        for (; i < 16; ++i) emf_mc_tri_table[c][i] = -1;
    }
    done = 1;
}
#endif
