"""bench.py contract (driver-facing): one JSON line with the agreed keys; smoke() runs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ); e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_json_contract():
    d = _run(["--steps", "1", "--warmup", "1", "--pairs-per-step", "8"])
    assert d["n_ranks_seen"] == 1 and d["config"]["distinct_pairs"] == 8 and d["roofline"]["traffic"] is None
    ex = d["extras"]
    for k in ("gicp_af_radius_rule", "fgr_plus_gicp", "config5_2M_points_5_scales_64nn"):
        assert k in ex and "error" not in ex[k], (k, ex.get(k))
        assert ex[k]["pairs_per_s"] > 0 and ex[k]["err_vs_planted"]["rad"] < 2e-3 and ex[k]["err_vs_planted"]["m"] < 2e-2, (k, ex[k])
    assert ex["fgr_plus_gicp"]["roofline"]["bound"] == "mfma" and ex["fgr_plus_gicp"]["roofline"]["achieved"] > 0
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]          # value = pairs of the K steps / their wall time
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "pairs/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 20 * c["value"]                                                  # north star: >= 20x the CPU path at 1 GPU
    assert d["config"]["err_vs_planted"]["rad"] < 2e-3 and d["config"]["err_vs_planted"]["m"] < 2e-2


def test_bench_two_rank_rehearsal_on_one_gpu():
    """The N > 1 path (sharding of the batches, gathered records, max-over-ranks time) with gloo on one device."""
    e = dict(os.environ, PCR_BENCH_REHEARSE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--pairs-per-step", "6", "--no-cpu-baseline"],
                         cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["gathered_records"] == 12 and "cpu_baseline" not in d


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` (no torchrun around it): bench.py starts the ranks itself; both share device 0 here."""
    d = _run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--pairs-per-step", "4", "--no-cpu-baseline"], env={"PCR_BENCH_REHEARSE": "1"})
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["config"]["gathered_records"] == 8


def test_bench_config3_two_rank_rehearsal_on_one_gpu():
    """`bench.py --gpus 2 --config3`: BASELINE config 3's block as a strong-scaling run -- two ranks (both on device 0, gloo) take cost-balanced
    blocks of the tiled golden NCLT pairs, each runs ONE fgr+gicp plan, one all-gather; every pose reaches the reference's shipped GICP pose."""
    d = _run(["--gpus", "2", "--config3", "--config3-tiles", "3", "--warmup", "1"], env={"PCR_BENCH_REHEARSE": "1"})
    c = d["config"]
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["dry_run"] is False
    assert c["pairs"] == 24 and c["gathered_records"] == 24 and c["tables_identical"] is True and c["poses_valid"] is True
    assert sum(r["pairs"] for r in c["per_rank"]) == 24 and all(r["wall_s"] > 0 and r["upload_s"] > 0 for r in c["per_rank"])
