"""Pins the CPU oracle against the reference's own shipped outputs:
multiscale GICP with script-2 parameters, started from the shipped FGR pose, must land on
the shipped FGR+GICP pose within the north-star tolerance (1e-4 rad / 1e-3 m).

Noise floor (DESIGN.md "Parity"): the L1-IRLS trajectory is chaotic (weights 1/|r| blow up on
near-zero whitened residuals), so the END pose of the reference algorithm itself scatters by up
to a few mm on some pairs when only the floating-point summation order changes.  The committed
golden pairs are members of the pinned list whose attractor is tight; pair 899 is kept as a small
and *noisy* example with a looser bound, and the full 40-pair sweep (needs /root/reference) asserts
the pass RATE.
"""
import os

import numpy as np
import pytest

from conftest import SCRIPT2_DISTS, SCRIPT2_VOXELS, pose_error, pkg

TOL_RAD, TOL_M = 1e-4, 1e-3          # BASELINE.json north_star
NOISY = {0, 899}                     # pairs whose L1 attractor is wide: see test_noisy_pairs_reach_the_shipped_pose_within_their_own_spread
NOISY_TOL_RAD, NOISY_TOL_M = 1e-3, 1e-2


def test_oracle_reproduces_shipped_gicp_pose(oracle, golden_pair):
    g = golden_pair
    r = oracle.multiscale_gicp(g["source"], g["target"], SCRIPT2_VOXELS, SCRIPT2_DISTS, g["T_fgr"])
    ang, dt = pose_error(r.transformation, g["T_gicp"])
    tr, tm = (NOISY_TOL_RAD, NOISY_TOL_M) if int(g["pair"]) in NOISY else (TOL_RAD, TOL_M)
    assert ang <= tr and dt <= tm, (int(g["pair"]), ang, dt)
    # the shipped FGR pose itself is far outside the tolerance, so the test is not vacuous
    a0, d0 = pose_error(g["T_fgr"], g["T_gicp"])
    assert a0 > 10 * TOL_RAD or d0 > 10 * TOL_M


def test_noisy_pairs_reach_the_shipped_pose_within_their_own_spread(oracle, golden_pair):
    """Pairs 0 and 899: changing ONLY the float64 summation chunking of the normal equations moves the oracle's end pose by up to
    1e-3 rad / 12 mm (on pair 0 the last scale stops after ~48 or after ~92 iterations depending on it).  The shipped pose is one
    sample of the same scatter: some chunking reproduces it inside the north-star tolerance, every chunking stays inside the loose
    bound, and the scatter is larger than the tolerance -- which is why the device is compared with the oracle through
    conftest.l1_tolerance on these pairs and not through a constant."""
    g = golden_pair
    if int(g["pair"]) not in NOISY:
        pytest.skip("tight attractor")
    errs = []
    for chunk in (256, 32, 64, 128, 512, 1024, 4096):
        old = oracle.set_sum_chunk(chunk)
        try:
            r = oracle.multiscale_gicp(g["source"], g["target"], SCRIPT2_VOXELS, SCRIPT2_DISTS, g["T_fgr"])
        finally:
            oracle.set_sum_chunk(old)
        errs.append(pose_error(r.transformation, g["T_gicp"]))
    errs = np.array(errs)
    assert ((errs[:, 0] <= TOL_RAD) & (errs[:, 1] <= TOL_M)).any(), errs           # the shipped pose IS reachable by the restated algorithm
    assert (errs[:, 0] <= 2 * NOISY_TOL_RAD).all() and (errs[:, 1] <= 2 * NOISY_TOL_M).all(), errs       # pair 899, chunk 32: 1e-3 rad / 12 mm
    assert errs[:, 1].max() - errs[:, 1].min() > 0.5 * TOL_M, errs                 # and the scatter is real


def test_oracle_is_deterministic_across_thread_counts(oracle, small_pair):
    g = small_pair
    prev = oracle.set_num_threads(1)
    try:
        r1 = oracle.multiscale_gicp(g["source"], g["target"], SCRIPT2_VOXELS[:2], SCRIPT2_DISTS[:2], g["T_fgr"], max_it=15)
        oracle.set_num_threads(3)
        r3 = oracle.multiscale_gicp(g["source"], g["target"], SCRIPT2_VOXELS[:2], SCRIPT2_DISTS[:2], g["T_fgr"], max_it=15)
    finally:
        oracle.set_num_threads(prev)
    assert np.array_equal(r1.transformation, r3.transformation)


def test_oracle_ablation_l2_loss_fails(oracle, golden_pair):
    """Sanity of the pin: the wrong robust kernel must NOT reproduce the shipped pose."""
    g = golden_pair
    if int(g["pair"]) != 10:
        pytest.skip("one pair is enough")
    r = oracle.multiscale_gicp(g["source"], g["target"], SCRIPT2_VOXELS, SCRIPT2_DISTS, g["T_fgr"], loss=oracle.LOSS_L2)
    ang, dt = pose_error(r.transformation, g["T_gicp"])
    assert dt > TOL_M or ang > TOL_RAD


PINNED = [0, 5, 10, 25, 45, 85, 100, 125, 145, 165, 200, 205, 245, 265, 405, 425, 445, 450, 465, 500, 505, 545, 565, 585,
          600, 605, 625, 665, 700, 705, 725, 745, 765, 800, 805, 825, 845, 865, 885, 899]


@pytest.mark.skipif(not os.path.isdir(os.environ.get("PCR_REFERENCE_DIR", "/root/reference")) or
                    not os.environ.get("PCR_FULL_PINNED"), reason="full pinned sweep needs /root/reference and PCR_FULL_PINNED=1")
def test_oracle_full_pinned_list(oracle):
    ref = os.environ.get("PCR_REFERENCE_DIR", "/root/reference")
    pio = pkg("io")
    errs = []
    for i in PINNED:
        s = pio.read_pcd_xyz(f"{ref}/nuvens/nuvens_pre_processadas/NCLT/s{i + 1}.pcd")
        t = pio.read_pcd_xyz(f"{ref}/nuvens/nuvens_pre_processadas/NCLT/s{i}.pcd")
        T0 = pio.read_pose(f"{ref}/relative_poses_FGR/NCLT/pose_{i + 1}_{i}.txt")
        Tg = pio.read_pose(f"{ref}/relative_poses_FGR_GICP/NCLT/pose_{i + 1}_{i}.txt")
        r = oracle.multiscale_gicp(s, t, SCRIPT2_VOXELS, SCRIPT2_DISTS, T0)
        errs.append(pose_error(r.transformation, Tg))
    errs = np.array(errs)
    ok = (errs[:, 0] <= TOL_RAD) & (errs[:, 1] <= TOL_M)
    assert ok.mean() >= 0.7, errs
    assert np.median(errs[:, 0]) <= TOL_RAD and np.median(errs[:, 1]) <= TOL_M
    assert errs[:, 0].max() <= 2e-3 and errs[:, 1].max() <= 3e-2
