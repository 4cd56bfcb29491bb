"""Circuit drivers: script-equivalent stages over files in the reference's on-disk formats."""
import os

import numpy as np
import pytest

import pcr_amd
from conftest import GOLDEN, pose_error

H = np.load(os.path.join(GOLDEN, "host_pose_algebra.npz"))


def test_stage3_on_shipped_facade_circuit(tmp_path):
    rel = list(H["Facade_relative"]); n = len(rel)
    d = tmp_path / "relative"; d.mkdir()
    for i, T in enumerate(rel):
        pcr_amd.io.write_pose(str(d / pcr_amd.io.relative_pose_name(i, n)), T)
    assert (d / f"pose_0_{n - 1}.txt").exists() and (d / "pose_1_0.txt").exists()       # closure under the name stage 2/3 read
    out = pcr_amd.drivers.stage3_refine(str(d), n, out_dir=str(tmp_path / "abs"))
    np.testing.assert_allclose(np.stack(out["LUM"]), H["Facade_lum"], atol=1e-9)          # reference's own LUM (unit weights)
    np.testing.assert_allclose(np.stack(out["plain"]), H["Facade_absolute"], atol=1e-12)
    np.testing.assert_allclose(out["closure"], H["Facade_closure"], atol=1e-12)
    for name in ("LUM", "SLERP", "SLERP_LUM"):
        back = [pcr_amd.io.read_pose(str(tmp_path / "abs" / name / f"pose{i}.txt")) for i in range(n)]
        np.testing.assert_allclose(np.stack(back), np.stack(out[name]), atol=1e-15)
    # ground-truth comparison path (script 3, step 9) against the plain composition written as "ground truth"
    gt = tmp_path / "gt"; gt.mkdir()
    for i, T in enumerate(out["plain"]):
        pcr_amd.io.write_pose(str(gt / f"pose{i}.txt"), T)
    err = pcr_amd.drivers.stage3_refine(str(d), n, groundtruth_dir=str(gt))["errors"]
    assert max(err["plain"][0]) < 1e-12 and max(err["plain"][1]) < 1e-12 and max(err["SLERP_LUM"][1]) > 1e-4


def test_cli_stage3(tmp_path, capsys):
    rel = list(H["Courtyard_relative"]); n = len(rel)
    for i, T in enumerate(rel):
        pcr_amd.io.write_pose(str(tmp_path / pcr_amd.io.relative_pose_name(i, n)), T)
    assert pcr_amd.drivers.main(["stage3", "--relative", str(tmp_path), "--n", str(n), "--out", str(tmp_path / "o")]) == 0
    assert "closure error" in capsys.readouterr().out
    assert (tmp_path / "o" / "SLERP_LUM" / f"pose{n - 1}.txt").exists()


@pytest.mark.gpu
def test_stage1_stage2_on_a_two_cloud_circuit(tmp_path):
    """Clouds s0 (target) and s1 (source) of golden pair 10 as a circuit of two: pair 0 = s1 -> s0, pair 1 closes the
    loop (s0 -> s1).  Stage 1 writes %.10f FGR poses under the names stage 2 reads; stage 2 refines them."""
    g = np.load(os.path.join(GOLDEN, "nclt_pair_010.npz"))
    clouds = tmp_path / "clouds"; clouds.mkdir()
    pcr_amd.io.write_pcd_xyz(str(clouds / "s0.pcd"), g["target"]); pcr_amd.io.write_pcd_xyz(str(clouds / "s1.pcd"), g["source"])
    fgr = pcr_amd.drivers.stage1_fgr(str(clouds), str(tmp_path / "fgr"), 2, voxel_size=0.1, inflight=2, seed=2024, verbose=False)
    assert sorted(os.listdir(tmp_path / "fgr")) == ["pose_0_1.txt", "pose_1_0.txt"]
    a, d = pose_error(fgr[0], g["T_fgr"]); assert a < 2e-2 and d < 0.4, (a, d)
    txt = open(tmp_path / "fgr" / "pose_1_0.txt").read().split()
    assert all(len(v.split(".")[1]) == 10 for v in txt)                                   # S1:177 fmt="%.10f"
    rel, ab, table = pcr_amd.drivers.stage2_mgicp(str(clouds), str(tmp_path / "fgr"), str(tmp_path / "gicp"), 2, n_scales=5, iterations=100,
                                                  inflight=2, absolute_dir=str(tmp_path / "abs"), verbose=False)
    a, d = pose_error(rel[0], g["T_gicp"]); assert a < 3e-4 and d < 3e-3, (a, d)
    a, d = pose_error(rel[1], np.linalg.inv(g["T_gicp"])); assert a < 2e-3 and d < 2e-2, (a, d)     # the way back
    assert [r["pair"] for r in table] == [0, 1] and all(r["fitness"] > 0.3 for r in table)
    assert np.allclose(ab[0], np.eye(4)) and np.allclose(ab[1], rel[0])
    assert sorted(os.listdir(tmp_path / "abs")) == ["pose0.txt", "pose1.txt"]
    back = pcr_amd.io.load_relative_poses(str(tmp_path / "gicp"), 2)
    np.testing.assert_allclose(np.stack(back), np.stack(rel), atol=1e-15)


@pytest.mark.gpu
def test_stage2_two_ranks_equal_one_rank_on_the_facade_circuit(tmp_path):
    """the shipped Facade circuit (7 clouds, 7 pairs incl. the loop closure) through `drivers stage2` as ONE
    process and as TWO ranks (contiguous blocks of 3 and 4 pairs, one all-gather of the pose records; both ranks on this box's one
    device, gloo): the pose files rank 0 writes must be the same bits.  Nothing is pinned: the drivers size their groups by the clouds
    (group=None), which makes the kernel forms a function of the pair alone (`pcr_pairs_plan.pair_forms`), whatever the cut -- with
    --inflight 2 one rank runs groups of 4 + 3, two ranks 2 + 1 and 2 + 2."""
    import subprocess
    import sys
    from conftest import ROOT
    g = np.load(os.path.join(GOLDEN, "facade_loop.npz"))
    n = 7
    clouds = tmp_path / "clouds"; clouds.mkdir(); init = tmp_path / "init"; init.mkdir()
    for i in range(n):
        pcr_amd.io.write_pcd_xyz(str(clouds / f"s{i}.pcd"), g[f"s{i}"])
        pcr_amd.io.write_pose(str(init / pcr_amd.io.relative_pose_name(i, n)), g["T_fgr"][i])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PCR_REHEARSE="1", PYTHONPATH=ROOT)
    script = tmp_path / "run_stage.py"          # (the package name has hyphens: importlib, not `-m`)
    # (and torchrun's own argument parser trips over `--n`: the stage arguments travel in the environment)
    script.write_text("import importlib, json, os, sys\nsys.exit(importlib.import_module('point-cloud-registration-with-global-refinement_amd.drivers').main(json.loads(os.environ['PCR_STAGE_ARGS'])))\n")
    common = ["stage2", "--clouds", str(clouds), "--init", str(init), "--n", str(n), "--scales", "3", "--iterations", "30", "--inflight", "2"]
    import json
    one = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=dict(env, PCR_STAGE_ARGS=json.dumps(common + ["--out", str(tmp_path / "one")])),
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                          str(script)], cwd=ROOT, env=dict(env, PCR_STAGE_ARGS=json.dumps(common + ["--out", str(tmp_path / "two")])),
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    a = pcr_amd.io.load_relative_poses(str(tmp_path / "one"), n); b = pcr_amd.io.load_relative_poses(str(tmp_path / "two"), n)
    assert np.array_equal(np.stack(a), np.stack(b))                                        # every pair, every bit
    assert sorted(os.listdir(tmp_path / "one")) == sorted(os.listdir(tmp_path / "two"))
    for i in range(n):                                                                      # and they are registrations, not garbage: near the shipped GICP poses
        ang, dt = pose_error(a[i], g["T_gicp"][i])
        assert ang < 2e-2 and dt < 0.2, (i, ang, dt)


@pytest.mark.gpu
def test_stage12_two_ranks_equal_one_rank_on_the_facade_circuit(tmp_path):
    """BASELINE config 3's per-rank building block -- FGR (script-1 parameters) + 5-scale MGICP (script-2 table) of every pair as ONE
    `fgr+gicp` plan per rank (`drivers stage12`) -- on the shipped Facade circuit as one process and as two ranks (cost-balanced blocks,
    both on this box's one device, gloo): the FGR and the GICP pose files must be the same bits, and they must be registrations
    (the refined poses close to the shipped ones)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    g = np.load(os.path.join(GOLDEN, "facade_loop.npz"))
    n = 7
    clouds = tmp_path / "clouds"; clouds.mkdir()
    for i in range(n):
        pcr_amd.io.write_pcd_xyz(str(clouds / f"s{i}.pcd"), g[f"s{i}"])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PCR_REHEARSE="1", PYTHONPATH=ROOT)
    script = tmp_path / "run_stage.py"
    script.write_text("import importlib, json, os, sys\nsys.exit(importlib.import_module('point-cloud-registration-with-global-refinement_amd.drivers').main(json.loads(os.environ['PCR_STAGE_ARGS'])))\n")
    common = ["stage12", "--clouds", str(clouds), "--n", str(n), "--scales", "5", "--iterations", "30", "--inflight", "2", "--seed", "11"]
    runs = {}
    for name, launcher in (("one", [sys.executable]), ("two", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                               "--master-port", "29541"])):
        args = common + ["--out", str(tmp_path / name / "gicp"), "--fgr-out", str(tmp_path / name / "fgr")]
        r = subprocess.run(launcher + [str(script)], cwd=ROOT, env=dict(env, PCR_STAGE_ARGS=json.dumps(args)), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[name] = (pcr_amd.io.load_relative_poses(str(tmp_path / name / "gicp"), n), pcr_amd.io.load_relative_poses(str(tmp_path / name / "fgr"), n))
    assert np.array_equal(np.stack(runs["one"][1]), np.stack(runs["two"][1]))          # FGR poses: same seeds per pair index whatever the shard
    assert np.array_equal(np.stack(runs["one"][0]), np.stack(runs["two"][0]))          # refined poses: every pair, every bit
    for i in range(n):
        ang, dt = pose_error(runs["one"][0][i], g["T_gicp"][i])
        assert ang < 2e-2 and dt < 0.2, (i, ang, dt)
