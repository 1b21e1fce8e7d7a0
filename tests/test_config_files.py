"""The reference's configuration files read without Boost (core/Config.cpp; reference apps/EM-Fusion.cpp:40-104 value
parsers, 268-371 option list + boost::program_options::parse_config_file; calibration.txt: 399-410)."""
from pathlib import Path

import pytest

from emfusion_amd import pipeline

CFG = """\
## a comment block like the licence header of the reference's files
[Params]
# frameSize should be given as two integers
frameSize = 320 240
[Params.intr]
fx = 262.5
fy = 263.0   # trailing comment
cx = 159.5
cy = 119.5

[Params]
bilateral_sigma_depth = 0.05
bilateral_kernel_size = 5
globalVolumeDims = 256, 256 128
globalVoxelSize = 0.015
globalRelTruncDist = 8.0
objVolumeDims = 96 96 96
volumePose = 0 0.5 0.96
maxTrackingIter = 40
maskRCNNFrames = 10
visibilityThresh = 400
boundary = 10
ignore_person = yes

[Params.tsdfParams]
tau = 1e3
eps1 = 1e-8
huberThresh = 0.1
maxTSDFWeight = 32.0
assocSigma = 0.03
alpha = 0.7
uniPrior = 0.5

[Params.MaskRCNNParams]
FILTER_CLASSES = person
STATIC_OBJECTS = traffic light
STATIC_OBJECTS = dining table
"""


def test_every_option_of_the_reference_is_read(tmp_path):
    (tmp_path / "a.cfg").write_text(CFG)
    prm, f = pipeline.load_config(tmp_path / "a.cfg")
    assert (prm.width, prm.height) == (320, 240)
    assert list(prm.K) == [262.5, 0, 159.5, 0, 263.0, 119.5, 0, 0, 1]
    assert list(prm.bg_res) == [256, 256, 128] and abs(prm.bg_voxel_size - 0.015) < 1e-9 and prm.bg_rel_truncdist == 8.0
    assert list(prm.obj_res) == [96, 96, 96] and [round(v, 6) for v in prm.volume_pose_t] == [0, 0.5, 0.96]
    assert (prm.max_tracking_iter, prm.mask_frames, prm.visibility_thresh, prm.boundary) == (40, 10, 400, 10)
    assert prm.max_tsdf_weight == 32.0 and abs(prm.assoc_sigma - 0.03) < 1e-8 and abs(prm.alpha - 0.7) < 1e-7 and prm.uni_prior == 0.5
    assert f["Params.bilateral_sigma_depth"] == "0.0500000007" and f["Params.bilateral_kernel_size"] == "5"
    assert f["Params.tsdfParams.huberThresh"] == "0.100000001" and f["Params.tsdfParams.tau"] == "1000"
    assert f["Params.ignore_person"] == "yes"
    assert f["Params.MaskRCNNParams.FILTER_CLASSES"] == ["person"]
    assert f["Params.MaskRCNNParams.STATIC_OBJECTS"] == ["traffic light", "dining table"]  # the whole value, blanks included
    # what the file does not name keeps the reference's default (data.h)
    assert f["Params.volPad"] == "2" and f["Params.matchIOUThresh"] == "0.200000003" and f["Params.tsdfParams.nu_init"] == "2"
    # usable as it is
    assert isinstance(prm, pipeline.FusionParams)


def test_defaults_without_a_file_are_the_reference_defaults():
    prm, f = pipeline.load_config()
    assert (prm.width, prm.height) == (640, 480) and list(prm.bg_res) == [512, 512, 512] and list(prm.obj_res) == [64, 64, 64]
    assert f["Params.visibilityThresh"] == "1600" and f["Params.ignore_person"] == "no"
    assert "Params.MaskRCNNParams.STATIC_OBJECTS" not in f


@pytest.mark.parametrize("text, message", [
    ("[Params]\nnoSuchKey = 1\n", "unrecognised option 'Params.noSuchKey'"),
    ("[Params]\nboundary = 10\nboundary = 12\n", "more than once"),
    ("[Params]\njust some words\n", "unrecognized line"),
    ("[Params]\nframeSize = 640\n", "two integers"),
    ("[Params]\nglobalVolumeDims = 1 2 x\n", "invalid option value"),
    ("[Params]\nignore_person = maybe\n", "invalid bool"),
    ("[Params]\nboundary = 1.5\n", "invalid option value"),
])
def test_what_the_reference_parser_rejects(tmp_path, text, message):
    (tmp_path / "bad.cfg").write_text(text)
    with pytest.raises(pipeline.FusionError, match=message):
        pipeline.load_config(tmp_path / "bad.cfg")
    with pytest.raises(pipeline.FusionError, match="can not read"):
        pipeline.load_config(tmp_path / "missing.cfg")


def test_calibration_file_of_the_cofusion_datasets(tmp_path):
    (tmp_path / "calibration.txt").write_text("528 527.5 320 240 640 480\n")
    prm, _ = pipeline.load_config(None, tmp_path / "calibration.txt")
    assert list(prm.K) == [528, 0, 320, 0, 527.5, 240, 0, 0, 1] and (prm.width, prm.height) == (640, 480)
    (tmp_path / "short.txt").write_text("500 500 100 90\n")  # intrinsics only
    prm, _ = pipeline.load_config(None, tmp_path / "short.txt")
    assert list(prm.K)[:6] == [500, 0, 100, 0, 500, 90] and (prm.width, prm.height) == (640, 480)
    prm, _ = pipeline.load_config(None, tmp_path / "none.txt")  # a missing file is ignored, as in the reference
    assert prm.K[0] == 525.0


REF = Path("/root/reference/config")


@pytest.mark.skipif(not REF.is_dir(), reason="the reference tree exists in the build container only")
def test_the_reference_s_own_files_load():
    """Read where they lie; nothing of them is kept here."""
    prm, f = pipeline.load_config(REF / "default.cfg")
    assert (prm.width, prm.height) == (640, 480) and prm.K[0] == 525.0 and prm.K[2] == 319.5
    assert list(prm.bg_res) == [512, 512, 512] and abs(prm.bg_voxel_size - 0.01) < 1e-9 and prm.visibility_thresh == 1600
    assert abs(prm.volume_pose_t[2] - 2.56) < 1e-6 and len(f["Params.MaskRCNNParams.STATIC_OBJECTS"]) == 13
    prm, f = pipeline.load_config(REF / "tum.cfg")
    assert f["Params.ignore_person"] == "yes" and f["Params.MaskRCNNParams.FILTER_CLASSES"] == ["person"]
    prm, f = pipeline.load_config(REF / "room4.cfg")
    assert abs(prm.bg_voxel_size - 0.015) < 1e-9 and abs(prm.volume_pose_t[2] - 0.96) < 1e-6
    prm, f = pipeline.load_config(REF / "co-fusion-real.cfg")
    assert f["Params.MaskRCNNParams.STATIC_OBJECTS"][-1] == "umbrella"
