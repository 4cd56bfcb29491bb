"""Lockstep groups of pairs (``pcr_pairs_plan.group``): the same per-pair arithmetic as the pair-by-pair path, batched over the
launches.  The pair loop itself is 2_MGICP...py:187-214; the reference has no grouping, so the checks are (i) bit identity with the
pair-by-pair path at equal summation grouping and (ii) the oracle on the golden pairs."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, pkg, pose_error, l1_tolerance

pytestmark = pytest.mark.gpu


def _helper(env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "group_pose.py")], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (env, out.stderr[-2000:])
    lines = out.stdout.splitlines()
    groups = [l for l in lines if l.startswith("GROUP ")]
    poses = {l.split()[0]: np.array([float(v) for v in l.split()[1:]]).reshape(-1, 4, 4) for l in lines if l.startswith("POSES")}
    return groups, poses


@pytest.mark.parametrize("rule", ["given", "af"])
def test_groups_are_bit_identical_at_equal_summation_grouping(rule):
    """Groups of 1, 2, 3 and 8 over 5 pairs of different sizes (ragged last group, group larger than the batch): poses, iteration
    counts, cloud counts, radii and correspondence sets are the same bits when the iteration kernel's tile (PCR_ICP_PPL) and the k-NN kernel
    (PCR_KNN_WAVE) are the same."""
    for ppl, wave in (("1", "0"), ("2", "1")):          # (one search kernel for all batch sizes: by default batches of 6 searches and more take the wavefront k-NN kernel)
        groups, _ = _helper({"PCR_ICP_PPL": ppl, "GROUP_POSE_RULE": rule, "PCR_KNN_WAVE": wave})
        assert len(groups) == 4
        assert all(g == groups[0] for g in groups), (ppl, groups)


def test_groups_are_bit_identical_by_default_however_the_batch_is_cut():
    """No switch pinned (round-3 advisor finding): a plan with lockstep groups runs EVERY unit with the group forms of the kernels -- a
    ragged last group of one pair too (5 pairs in groups of 2 are 2 + 2 + 1, in groups of 4 they are 4 + 1) -- and with
    `pair_forms` (what group=None sets) also a pair registered pair by pair: same bits for every cut."""
    groups, _ = _helper({"GROUP_POSE_SIZES": "2,3,4,8"})
    assert len(groups) == 4
    assert all(g == groups[0] for g in groups), groups
    forms, _ = _helper({"GROUP_POSE_SIZES": "1,2,0", "GROUP_POSE_PAIR_FORMS": "1"})
    assert len(forms) == 3
    assert all(g == groups[0] for g in forms), (groups[0], forms)


def test_all_scales_in_one_loop_gives_the_scale_by_scale_bits():
    """NCLT-size groups run all scales of a pair behind one argument slot: the last workgroup of the launch in which a scale's criteria hold moves
    the pair on to its next scale on the device (pcr_dev_gicp_group_scales), so a pair does not wait for the slowest pair of every scale.
    PCR_ICP_SCALES=0 runs the scales one after the other (pcr_dev_gicp_group per scale): poses, iteration counts, cloud counts, radii and
    correspondence sets are the same bits, for every cut of the batch."""
    a, _ = _helper({"GROUP_POSE_SIZES": "2,3,8", "PCR_ICP_SCALES": "1"})
    b, _ = _helper({"GROUP_POSE_SIZES": "2,3,8", "PCR_ICP_SCALES": "0"})
    assert len(a) == len(b) == 3
    assert all(g == a[0] for g in a) and all(g == a[0] for g in b), (a, b)


def test_fgr_plus_gicp_groups_are_bit_identical():
    """Stage FGR + GICP (Coarse_to_fine / full_registration, ALL_FUNCTIONS.py:317-332, 349-392) with lockstep groups: registro_FGR runs
    pair by pair, the GICP of the group in lockstep from the FGR poses with the FGR normals as orientation prior and the AF radius rule,
    then the information matrices -- the same bits as pair by pair when the iteration kernel's tile is the same."""
    groups, _ = _helper({"PCR_ICP_TILE": "512", "GROUP_POSE_STAGE": "fgr+gicp", "GROUP_POSE_RULE": "af", "PCR_KNN_WAVE": "0"})
    assert len(groups) == 4
    assert all(g == groups[0] for g in groups), groups


def test_argument_forms_of_a_group_are_bit_identical():
    """Groups of up to 8 pairs hand their argument structs to the fused kernel BY VALUE (k_icp_fused_b), larger groups and
    PCR_ICP_BYVAL=0 through a device pointer (k_icp_fused_g); the cached graphs are patched with the new arguments from call to call:
    same bits either way."""
    a, _ = _helper({"PCR_ICP_PPL": "2"})
    b, _ = _helper({"PCR_ICP_PPL": "2", "PCR_ICP_BYVAL": "0"})
    assert a == b


def test_default_groups_agree_with_pair_by_pair():
    """Default policy (two source points per lane and the wavefront k-NN kernel inside groups, one point per lane and the octet kernel
    outside) only regroups float64 sums: smooth loss, 1e-7."""
    _, poses = _helper({"GROUP_POSE_LOSS": "l2"})
    for a, b in zip(poses["POSES1"], poses["POSES3"]):
        ang, d = pose_error(a, b)
        assert ang < 1e-7 and d < 1e-6, (ang, d)


def test_grouped_pairs_against_the_oracle(oracle, golden_pair_list):
    """Script-2 configuration (L1) in one lockstep group against the CPU oracle, tolerance derived from the oracle's own spread."""
    P = pkg(); reg = P.registration
    vox = P.script2.create_scales(5); dst = P.script2.max_correspondence_distances(vox)
    work = [(P.PointCloud(g["source"]), P.PointCloud(g["target"]), g["T_fgr"]) for g in golden_pair_list]
    rs = reg.register_pairs_plan(work, "gicp", vox, dst, reg.TransformationEstimationForGeneralizedICP(reg.L1Loss()), reg.ICPConvergenceCriteria(1e-6, 1e-6, 100),
                                 inflight=1, group=len(work))
    for g, r in list(zip(golden_pair_list, rs))[::3]:          # every third pair against the oracle (its spread runs take 10 s per pair on the CPU)
        ref, tol_r, tol_t, _ = l1_tolerance(oracle, lambda: oracle.multiscale_gicp(g["source"], g["target"], vox, dst, g["T_fgr"]), chunks=(64, 512, 4096))
        for a, b in zip(r.scales, ref.extra["scales"]):
            assert a["n_voxel"] == tuple(b["n_voxel"]) and a["n_clean"] == tuple(b["n_clean"])
        ang, d = pose_error(r.transformation, ref.transformation)
        assert ang <= tol_r and d <= tol_t, (int(g["pair"]), ang, d, tol_r, tol_t)


def test_a_bad_pair_in_a_group_fails_alone():
    """An argument error (NaN initial pose) inside a lockstep group lands on that pair only, like pair by pair; pairs the grouped
    path does not take (an empty cloud) run pair by pair with the same outcome."""
    P = pkg(); reg = P.registration
    rng = np.random.default_rng(3)
    good = rng.uniform(-5, 5, (4000, 3))
    bad = np.eye(4); bad[0, 0] = np.nan
    pairs = [(P.PointCloud(good), P.PointCloud(good + [0.05, 0, 0]), np.eye(4)), (P.PointCloud(good), P.PointCloud(good), bad)]
    for group in (1, 2):
        with pytest.raises(RuntimeError) as e:
            reg.register_pairs_plan(pairs, "gicp", [1.0, 0.5], [2.0, 1.0], inflight=1, group=group)
        assert "pair 1" in str(e.value), str(e.value)
    pairs[1] = (P.PointCloud(np.zeros((0, 3))), P.PointCloud(good), np.eye(4))
    outs = []
    for group in (1, 2):
        try:
            rs = reg.register_pairs_plan(pairs, "gicp", [1.0, 0.5], [2.0, 1.0], inflight=1, group=group)
            outs.append(("ok", rs[0].transformation, rs[1].fitness))
        except RuntimeError as e:
            outs.append(("error", str(e)))
    assert outs[0][0] == outs[1][0] and outs[0][2:] == outs[1][2:]
    if outs[0][0] == "ok":      # the group plan runs every unit with the group forms of the kernels (other summation grouping): smooth loss, rounding only
        ang, d = pose_error(outs[0][1], outs[1][1])
        assert ang < 1e-7 and d < 1e-6, (ang, d)


def test_group_none_sizes_groups_by_the_clouds(golden_pair_list):
    """``group=None`` (what the stage-2 driver passes) picks ``registration.default_group`` pairs per group: on NCLT-size pairs one group holds all of them;
    smooth loss: the poses agree with pair by pair to rounding."""
    P = pkg(); reg = P.registration
    vox = P.script2.create_scales(3); dst = P.script2.max_correspondence_distances(vox)
    work = [(P.PointCloud(g["source"]), P.PointCloud(g["target"]), g["T_fgr"]) for g in golden_pair_list[:4]]
    est = reg.TransformationEstimationForGeneralizedICP(reg.L2Loss()); crit = reg.ICPConvergenceCriteria(1e-6, 1e-6, 50)
    a = reg.register_pairs_plan(work, "gicp", vox, dst, est, crit, inflight=2, group=None)
    b = reg.register_pairs_plan(work, "gicp", vox, dst, est, crit, inflight=2, group=1)
    for x, y in zip(a, b):
        ang, d = pose_error(x.transformation, y.transformation)
        assert ang < 1e-7 and d < 1e-6, (ang, d)
        assert [s["n_clean"] for s in x.scales] == [s["n_clean"] for s in y.scales]
