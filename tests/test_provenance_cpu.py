"""What a bench line says about the evidence it cites: the committed counter-traffic files record the sha of the kernel
sources they were measured on, and `bench.stale` compares it with the sources of the running build — per kernel family
where the file records that (a change to the smoothing kernels does not age the tile kernel's figure)."""
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))

from tmc2rs import provenance  # noqa: E402


def test_family_hashes_cover_their_sources_only():
    all_sha = provenance.kernel_source_sha16()
    tiles, smooth = provenance.kernel_source_sha16("k_recon_tiles"), provenance.kernel_source_sha16("k_smooth")
    assert len({all_sha, tiles, smooth}) == 3 and all(len(x) == 16 for x in (all_sha, tiles, smooth))
    assert "vpcc_smooth.hip" not in provenance.FAMILY_SOURCES["k_recon_tiles"]
    assert "vpcc_tiles.hip" not in provenance.FAMILY_SOURCES["k_smooth"]
    for fam in provenance.FAMILY_SOURCES.values():
        assert set(fam) <= set(provenance.KERNEL_SOURCES)


def test_stale_prefers_the_family_hash():
    import bench
    fam = provenance.kernel_source_sha16("k_recon_tiles")
    assert bench.stale({"kernel_source_sha16": "0" * 16, "family_source_sha16": fam}, "k_recon_tiles") is False
    assert bench.stale({"kernel_source_sha16": provenance.kernel_source_sha16(), "family_source_sha16": "0" * 16}, "k_recon_tiles") is True
    assert bench.stale({"kernel_source_sha16": provenance.kernel_source_sha16()}, "k_smooth") is False
    assert bench.stale({"kernel_source_sha16": "0" * 16}, "k_smooth") is True


def test_committed_traffic_files_belong_to_the_committed_kernels():
    """The newest round's traffic files must have been measured on the sources in the tree (the round is not done
    while a kernel changed after its counter passes)."""
    import bench
    rounds = sorted(glob.glob(os.path.join(REPO, "profiles", "r*")))
    files = sorted(glob.glob(os.path.join(rounds[-1], "traffic*.json")))
    assert files
    for f in files:
        d = json.load(open(f))
        family = "k_smooth" if "k_smooth" in d["kernel"] else "k_recon_tiles"
        assert not bench.stale(d, family), f
