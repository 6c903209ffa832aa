"""The host simulator runs the GPU algorithm's per-block code (trp_lane.h) with sequential scans;
it must reproduce the CPU oracle bit for bit.  This is the CPU-side check of the bitmask
formulation (no GPU needed)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import sim_lib as S
import textgen
from parity import assert_same

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden_records():
    with open(os.path.join(G, "denoiser.json"), encoding="utf-8") as f:
        den = json.load(f)
    recs = [c["input"].encode("utf-8") for c in den["exact"] + den["property"]]
    recs += [s.encode() for s in den["table_line_true"] + den["table_line_false"]]
    return recs


def test_alpha_range_shortcut_matches_table():
    assert S.lib().trp_sim_check_alpha_shortcut() == 0


def test_plane_rules_match_table_for_every_code_point():
    assert S.lib().trp_sim_check_classify_codepoints() == 0


def test_monoids_are_associative():
    assert S.lib().trp_sim_check_monoids(12345, 200000) == 0


@pytest.mark.parametrize("tile_blocks", [1, 2, 3, 256])
@pytest.mark.parametrize("strip_md", [True, False])
def test_golden_inputs_as_one_batch(tile_blocks, strip_md):
    recs = _golden_records()
    cfg = O.make_config(True, 0.35, strip_md, 8, (2,), 4)
    data, offs = O.pack(recs)
    assert_same(O.batch(data, offs, cfg), S.batch(data, offs, cfg, tile_blocks), recs, "golden")


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("tile_blocks", [1, 4, 256])
def test_fuzz_enabled(seed, tile_blocks):
    recs = textgen.corpus(seed, 300)
    cfg = O.make_config(True, [0.35, 0.2, 0.5, 0.0, 1.0][seed % 5], seed % 3 != 0,
                        [1024, 16, 5, 0, 1][seed % 5], [(64,), (2, 3), (0,), (), (7, 1, 100)][seed % 5],
                        [512, 4, 0, 3, 1][(seed // 2) % 5])
    data, offs = O.pack(recs)
    assert_same(O.batch(data, offs, cfg), S.batch(data, offs, cfg, tile_blocks), recs, "fuzz%d" % seed)


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_disabled(seed):
    recs = textgen.corpus(100 + seed, 300)
    cfg = O.make_config(False, 0.35, True, [1024, 7][seed % 2], (64, 3), 5)
    data, offs = O.pack(recs)
    assert_same(O.batch(data, offs, cfg), S.batch(data, offs, cfg, 4), recs, "dis%d" % seed)


def test_edge_batches():
    cfg = O.make_config(True, 0.35, True, 4, (1,), 2)
    for recs in ([], [b""], [b"", b"", b""], [b"\n"], [b"a"], [b"a", b"", b"b\r", b"\nc"],
                 [b"x" * 64, b"y" * 64], [b"word " * 100 + b"\r", b"\n42 42 42 a"],
                 [b"|" * 200], [b" " * 130 + b"a" + b" " * 130], [b"| a |" + b" " * 70 + b"| b |"],
                 ["a　".encode() * 40], [("1 " * 40 + "w ").encode() * 5]):
        data, offs = O.pack(recs)
        for tb in (1, 256):
            assert_same(O.batch(data, offs, cfg), S.batch(data, offs, cfg, tb), recs, repr(recs)[:60])


@pytest.mark.parametrize("seed", range(6))
def test_scripts_and_rule_boundaries(seed):
    recs = textgen.script_corpus(700 + seed, 150)
    data, offs = O.pack(recs)
    for enabled, ratio, md in ((True, 0.35, True), (True, 0.8, False), (False, 0.35, True)):
        cfg = O.make_config(enabled, ratio, md, 12, (3,), 4)
        assert_same(O.batch(data, offs, cfg), S.batch(data, offs, cfg, [1, 4, 256][seed % 3]), recs, "scripts%d" % seed)
