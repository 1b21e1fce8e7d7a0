// Config.hpp -- the reference's configuration files (config/default.cfg, tum.cfg, room4.cfg, co-fusion-real.cfg) read
// into emf::Params, as apps/EM-Fusion.cpp:268-371 does with boost::program_options::parse_config_file: INI sections
// ("[Params.tsdfParams]" prefixes the keys below it), "key = value" lines, '#' starts a comment, the options of
// EM-Fusion.cpp:272-363 and no others (an unknown key is an error, as in the reference), multi-valued keys
// (frameSize, globalVolumeDims, objVolumeDims, volumePose: numbers split at ", ", EM-Fusion.cpp:40-104; volumePose is
// a translation), repeated keys for the two class lists.  And <dir>/calibration.txt of the Co-Fusion datasets
// ("fx fy cx cy width height", EM-Fusion.cpp:399-410).  Host-only code, no Boost.
#pragma once

#include <string>

#include "data.hpp"

namespace emf {

/** Overrides the fields of `p` the file names; throws std::runtime_error (with file and line) on anything the
 *  reference's parser would reject. */
void loadConfigFile(Params& p, const std::string& path);
/** fx fy cx cy [width height] -> p.intr / p.frameSize; false if the file does not exist (the reference ignores that). */
bool loadCalibrationFile(Params& p, const std::string& path);
/** Every configurable field as "Section.key = value" lines in the order of EM-Fusion.cpp:272-363 (for tests / logs). */
std::string dumpConfig(const Params& p);

}  // namespace emf
