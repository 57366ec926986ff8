"""The one stdout line of bench.py must stay parseable by the driver (which keeps ~8 000 characters of stdout): CPU checks
of compact_line() on canned records, and -- under -m gpu -- a real short run whose LAST stdout line is checked."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _canned():
    """the 30 KB record of round 4 (the line the driver could not keep), with the one key this round renamed"""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r4_bench_line.json")))
    rec["roofline"]["flat_list_equiv"]["required_over_peak"] = rec["roofline"]["flat_list_equiv"].pop("frac")
    return rec


def test_compact_line_of_the_round4_record_is_short_and_complete():
    import bench
    rec = _canned()
    assert len(json.dumps(rec)) > 20000                      # what went wrong in round 4
    txt = bench.compact_line(rec)
    assert len(txt) <= 4096 and "\n" not in txt
    out = json.loads(txt)
    for k in REQUIRED:
        assert k in out, k
    assert out["value"] == pytest.approx(rec["value"], rel=1e-5) and out["ms_per_step"] == pytest.approx(rec["ms_per_step"], rel=1e-5)
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "alg_bytes_per_launch", "binding"):
        assert k in rf, k
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    assert "frac" not in json.dumps(rf.get("flat_list_required_over_peak"))           # a plain number, not an object with a `frac`
    assert set(("achieved", "peak", "frac")) <= set(out["aes_roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(out["cpu_baseline"])
    assert out["config"]["workload"].startswith("d=500 CGD-15 64-bit")
    p12 = out["phase12"]
    assert set(p12) == {"c1", "c2", "c3-ti", "c3-ot", "c4", "all_exact"} and p12["all_exact"] is True
    assert all(isinstance(p12[k], float) for k in p12 if k != "all_exact")
    assert out["two_process_ring"]["seconds_garble_eval"] > 0
    sw = out["sweep64"]
    assert sw["exact"] is True and sw["seconds"] > 0 and set(sw["predicted_seconds_by_n_gpus"]) == {"2", "4", "8"}
    assert "timeline" not in txt and "model" not in sw


def test_compact_line_never_exceeds_the_limit_whatever_the_record_holds():
    import bench
    rec = _canned()
    rec["cpu_baseline"]["sample"] = "x" * 50000
    rec["cpu_baseline"]["model"] = "y" * 3000
    rec["devices"] = ["node%04d/%d" % (k, k) for k in range(400)]
    for res in rec["phase12"]:
        res["error"] = "e" * 9000
    txt = bench.compact_line(rec)
    assert len(txt) <= 4096
    out = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "roofline"):
        assert k in out, k


def test_compact_line_of_a_multi_gpu_record_carries_every_phase_and_the_prediction():
    import bench
    rec = _canned()
    rec.update(n_gpus=8, rccl_ranks=8, barrier_backend="nccl", devices=["box/%d" % k for k in range(8)], cpu_baseline=None,
               phase12=None, two_process_ring=None)
    sw = rec["sweep64"]
    pred = sw.pop("model")["by_n_gpus"]["8"]["predicted_seconds"]
    sw.update(n_gpus=8, block_lambdas=8, predicted_seconds=pred, measured_over_predicted=sw["seconds"] / pred,
              predicted_from="profiles/sweep_model.json", prefix_bytes_broadcast=116021248)
    out = json.loads(bench.compact_line(rec))
    assert out["rccl_ranks"] == 8 and len(out["devices"]) == 8
    for k in ("seconds", "create_s", "prefix_garble_s", "broadcast_s", "block_s", "gather_s", "predicted_seconds",
              "measured_over_predicted", "block_lambdas"):
        assert k in out["sweep64"], k


def test_committed_sweep_model_answers_for_2_4_8_gpus():
    import bench
    for n in (2, 4, 8):
        pred, src = bench.load_sweep_prediction(n, 64, 100, 15)
        assert pred is not None and pred > 0 and src
    assert bench.load_sweep_prediction(3, 64, 100, 15)[0] is None          # no such partition modelled: no number, no guess


@pytest.mark.gpu
def test_bench_last_stdout_line_parses_and_is_short(tmp_path):
    """the default bench shape (every leg but config 4's end-to-end run), one step: stdout is ONE line, the line is short"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LGC_BENCH_DETAIL_DIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LGC_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-c4", "--no-sweep-model", "--cpu-seconds", "3"]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = r.stdout.decode().rstrip("\n").splitlines()
    assert len(lines) == 1, [l[:80] for l in lines]
    assert len(lines[0]) <= 4096
    out = json.loads(lines[0])
    for k in REQUIRED + ("exact_vs_oracle", "aes_roofline", "phase12", "two_process_ring", "sweep64"):
        assert k in out, k
    assert out["exact_vs_oracle"] is True and out["value"] > 1e9 and out["n_gpus"] == 1
    assert 0 < out["roofline"]["frac"] < 1 and out["roofline"]["traffic"] is not None
    assert out["cpu_baseline"]["value"] > 0 and out["phase12"]["all_exact"] is True
    detail = json.load(open(os.path.join(str(tmp_path), out["detail"])))
    assert detail["phase12"][0]["timeline"]["steps"] and detail["roofline"]["flat_list_equiv"]["required_over_peak"] > 0
