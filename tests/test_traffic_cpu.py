"""tmc2rs/traffic.py (byte accounting behind bench.py's roofline fields) against a pixel-level restatement."""
import numpy as np

from tmc2rs import synth, traffic


def _pixel_level(fr):
    R, prec = fr["occupancy_resolution"], fr["occupancy_precision"]
    W, H, M = fr["width"], fr["height"], fr["map_count"]
    cover = traffic.cover_map(fr)
    covered = np.kron(cover >= 0, np.ones((R, R), dtype=bool))
    occ = np.kron(fr["occupancy"] != 0, np.ones((prec, prec), dtype=bool))[:H, :W]
    need = covered & occ
    out = {"pixels": int(need.sum()),
           "block_bytes": int(need.reshape(H // R, R, W // R, R).any(axis=(1, 3)).sum()) * R * R * 5 * M}
    for seg in (32, 64, 128):
        lp = seg // 2
        luma = need.reshape(H, W // lp, lp).any(axis=2).sum()
        ch = need.reshape(H // 2, 2, W // 2, 2).any(axis=(1, 3))
        chroma = ch.reshape(H // 2, (W // 2) // lp, lp).any(axis=2).sum()
        out["seg%d" % seg] = int(luma) * seg * 2 * M + int(chroma) * seg * 2 * M
    return out


def test_accounting_matches_pixel_level_restatement():
    for fr in (synth.make_frame(256, 192, 4, 16, seed=0xACC0, max_side=5, cover_target=0.5, size_skew=2.0),
               synth.make_frame(512, 256, 2, 16, seed=0xACC1, max_side=6, cover_target=0.4, size_skew=2.0)):
        fast, slow = traffic.frame_read_bytes(fr), _pixel_level(fr)
        for k, v in slow.items():
            assert fast[k] == v, k
        assert fast["seg32"] <= fast["seg64"] <= fast["seg128"]
        assert fast["pixel_bytes"] <= fast["seg32"]
