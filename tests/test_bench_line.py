"""bench.py's one-line record must stay parseable by the driver (round-4 verdict: a 37 KB line overflowed the driver's retained tail and
`BENCH_r04.parsed` was null).  The line is built by bench.compact_line from the full result object; everything else goes to
bench_detail.json / stderr.  No GPU: canned result objects, including the last full line a GPU visit produced (profiles/)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")
ROOFLINE = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms")
CPU = ("value", "unit", "cores", "kind", "sample")


def _leg(frames=48, ft=5, fps=(4.54, 5.71, 8.21)):
    """A tools/encoder_bench.run_config result with the per-leg diagnostics a real run carries (and then some)."""
    note = {"motion_estimate_calls": 123456789, "calls_with_lookup_context": 99999999, "lookups_served": 88888888, "lookup_hit_rate": 0.9293,
            "bytes_downloaded": 13300000000, "prose": "x" * 3000,
            "subpel_seam": {"subpel_compares_served": 7777777, "bytes_downloaded": 9400000000, "satd_lookups_served": 5555555, "prose": "y" * 2000},
            "lookahead_seam": {"frame_cost_estimates_served": 123, "prose": "z" * 2000}}
    return {"config": "configs[2]: 2160p 8-bit --preset slow --me star " + "w" * 300, "size": "3840x2160", "depth": 8, "preset": "slow",
            "options": {"pools": "16", "frame-threads": str(ft), "crf": "28"}, "pool_threads": 16, "reference_build": "b" * 200,
            "c": {"frames": frames, "seconds": 10.5, "fps": fps[0], "bytes": 1 << 20, "md5": "0" * 32},
            "csplit": {"frames": frames, "fps": fps[1], "md5": "0" * 32, "md5_equal_to_c_table": True},
            "seam": {"frames": frames, "fps": fps[2], "md5": "0" * 32, "md5_equal_to_c_table": True, "seam": note}}


def _canned(world=1):
    enc = {k: _leg() for k in ("cfg3", "cfg3f", "cfg4", "cfg5", "cfg2", "cfg3_v3", "cfg4_v3")}
    enc["cfg9"] = {"error": "RuntimeError('" + "e" * 500 + "')"}
    out = {"metric": "encoded fps + bit-exact check, 4K preset=slow, 1/2/4/8 MI355X vs host AVX2", "value": 552.035, "unit": "frames/s", "n_gpus": world,
           "steps": 200, "warmup": 5, "ms_per_step": 1.8115, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "W" * 900, "workload_detail": "D" * 3000, "parallelism": "P" * 600, "parallelism_detail": "Q" * 900, "ctus_per_frame": 2040,
                      "checksum": {"recon_y": 1 << 40, "recon_cb": 1 << 39, "recon_cr": 1 << 38}, "frames_per_step_per_gpu": 1, "sharding": "ring" if world > 1 else "none"},
           "stages_ms": {"stage_%02d_with_a_long_name_that_goes_on_and_on" % i: 0.1234 for i in range(40)},
           "roofline": {"bound": "hbm", "kernel": "me_ctu_q_kernel<best>", "achieved": 1948.79, "peak": 8000.0, "unit": "GB/s", "frac": 0.2436, "traffic": 193653240,
                        "traffic_source": "s" * 400, "frac_traffic": 0.017, "algorithmic_bytes_per_launch": 2768255520, "output_bytes_per_launch": 1387200,
                        "launch_ms": 1.4205, "valu": {"bound": "valu", "instruction": "v_qsad_pk_u16_u8", "pixel_candidates": 111470000000, "floor_ms": 0.7324,
                                                      "frac": 0.5156, "source": "t" * 300}, "note": "n" * 900},
           "stages_roofline": {"source": "r" * 300, "kernels": [{"kernel": "k%d" % i, "hbm_bytes": 1 << 30, "avg_us": 100.0, "frac_traffic": 0.1} for i in range(40)]},
           "cpu_baseline": {"value": 0.2796, "unit": "frames/s", "cores": 16, "kind": "port", "one_thread_value": 0.01778, "sample": "S" * 700, "sample_detail": "T" * 900},
           "bit_exact": True, "bit_exact_detail": {"ok": True, "stages": {"stage%d" % i: "equal" for i in range(30)}, "values_compared": 39343824, "what": "u" * 600},
           "encoder": enc}
    if world > 1:
        out["config"]["ring"] = {"ranks_seen": world, "transport": "abi", "bands_per_frame": 17, "refs": 1, "communicators": world, "comm_init_s": 1.234,
                                 "band_wait_ms_per_frame_max_over_ranks": 0.4321, "note": "v" * 600}
        out["config"]["band_rows"] = 2
        out["replicas"] = {"value": 4321.0, "unit": "frames/s", "ms_per_step": 1.85, "what": "g" * 300}
    out["encoder_summary"] = B.encoder_summary(enc)
    return out


@pytest.mark.parametrize("world", [1, 8])
def test_line_is_compact_and_carries_the_contract(world):
    out = _canned(world)
    line = json.dumps(B.compact_line(out), separators=(",", ":"))
    assert len(line) < B.MAX_LINE_BYTES == 4096, len(line)
    assert "\n" not in line
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert isinstance(d["value"], float) and isinstance(d["ms_per_step"], float) and d["n_gpus"] == world
    assert len(d["config"]["workload"]) <= 200 and d["config"]["ctus_per_frame"] == 2040
    for k in ROOFLINE:
        assert k in d["roofline"], k
    assert d["roofline"]["valu"] == {"floor_ms": 0.7324, "frac": 0.5156}
    for k in CPU:
        assert k in d["cpu_baseline"], k
    assert d["bit_exact"] is True and d["bit_exact_values"] == 39343824
    # numbers only per configuration: C / control / seam frames per second, md5 equality, hit rate, GB downloaded
    s = d["encoder_summary"]
    for key in ("cfg3", "cfg3f", "cfg4", "cfg5", "cfg2", "cfg3_v3", "cfg4_v3"):
        leg = s[key]
        assert (leg["c_fps"], leg["control_fps"], leg["seam_fps"]) == (4.54, 5.71, 8.21) and leg["md5_equal"] is True
        assert leg["x_control"] == round(8.21 / 5.71, 3) and leg["hit_rate"] == 0.9293 and leg["gb_down"] == 22.7 and leg["satd_served"] == 5555555
        assert all(not isinstance(v, (str, dict, list)) for v in leg.values())
    assert "cfg9" not in s                       # a failed leg has no numbers; its message stays in the detail file
    if world > 1:
        r = d["config"]["ring"]
        assert r["ranks_seen"] == world and r["transport"] == "abi" and r["communicators"] == world and r["comm_init_s"] == 1.234
        assert r["band_wait_ms_per_frame_max_over_ranks"] == 0.4321 and "note" not in r
        assert d["replicas"] == {"value": 4321.0, "ms_per_step": 1.85, "unit": "frames/s"}      # ring and replicas side by side
    # nothing of the per-leg diagnostics, the stage comparison or the per-kernel table leaks into the line
    for banned in ("stages_roofline", "bit_exact_detail", "encoder", "workload_detail", "checksum"):
        assert banned not in d and banned not in d["config"]


def test_minimal_line_is_the_fallback():
    out = _canned(8)
    line = json.dumps(B.compact_line(out, minimal=True), separators=(",", ":"))
    assert len(line) < 2048
    d = json.loads(line)
    for k in CONTRACT + ("roofline", "cpu_baseline", "bit_exact"):
        assert k in d, k
    assert "encoder_summary" not in d and "stages_ms" not in d


def test_last_gpu_visits_full_line_compacts():
    """The 37 KB line of round 4's closing visit (profiles/r04_bench_final.json) through the same function."""
    p = os.path.join(ROOT, "profiles", "r04_bench_final.json")
    if not os.path.exists(p):
        pytest.skip("no committed full line")
    out = json.load(open(p))
    assert len(json.dumps(out)) > 30000
    out["encoder_summary"] = B.encoder_summary(out.get("encoder", {}))
    line = json.dumps(B.compact_line(out), separators=(",", ":"))
    assert len(line) < B.MAX_LINE_BYTES
    d = json.loads(line)
    assert d["value"] == out["value"] and d["ms_per_step"] == out["ms_per_step"] and d["roofline"]["frac"] == out["roofline"]["frac"]
    assert d["encoder_summary"]["cfg3"]["c_fps"] == out["encoder"]["cfg3"]["c"]["fps"]


def test_detail_file_holds_the_rest(tmp_path, monkeypatch, capsys):
    out = _canned(1)
    monkeypatch.setenv("X265HIP_BENCH_DETAIL", str(tmp_path / "detail.json"))
    B.write_detail(out, None)
    back = json.load(open(tmp_path / "detail.json"))
    assert back["bit_exact_detail"]["values_compared"] == 39343824 and "cfg3" in back["encoder"] and "stages_roofline" in back
    cap = capsys.readouterr()
    assert cap.out == ""                          # stdout belongs to the one line
    assert "bench.py detail" in cap.err


def test_encoder_plan_names_every_default_configuration():
    class A:
        encoder, encoder_tables, encoder_frames, encoder_frame_threads = "cfg3,cfg3f,cfg4,cfg5,cfg2", "c,csplit,seam", 0, 0
    plan = B.encoder_plan(A)
    assert [p["name"] for p in plan] == ["cfg3", "cfg3f", "cfg4", "cfg5", "cfg2", "cfg3_v3", "cfg4_v3"]
    by = {p["name"]: p for p in plan}
    assert by["cfg3"]["frames"] == 48 and by["cfg3"]["frame_threads"] == 5 and by["cfg5"]["frames"] == 3
    assert by["cfg4_v3"]["build"] == "v3" and by["cfg4"]["seam"]["slots"] == 40 and by["cfg3"]["seam"]["slots"] == 24
    # configured by the one-box matrix (profiles/r05_seam_matrix.txt): no SAD lookups at 8 bits, the 32x32-and-up rasters above
    assert by["cfg3"]["seam"]["no_sad"] and by["cfg3f"]["seam"]["no_sad"] and not by["cfg4"]["seam"]["no_sad"] and by["cfg4"]["seam"]["min_level"] == 2
    assert by["cfg3_v3"]["tables"][:2] == ["c", "csse"]
    assert len(json.dumps(plan)) < 100000         # goes through argv
