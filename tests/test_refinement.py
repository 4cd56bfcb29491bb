"""Host-side global refinement: stub-import goldens for the numpy-only reference functions,
property tests for the SLERP family (no runnable oracle: numpy-quaternion is absent)."""
import os

import numpy as np
import pytest

import pcr_amd
from pcr_amd import refinement as rf

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "host_pose_algebra.npz"))
DATASETS = ("Facade", "Courtyard")


def _rel(ds):
    return [T for T in G[f"{ds}_relative"]]


@pytest.mark.parametrize("ds", DATASETS)
def test_pose_algebra_matches_reference_goldens(ds):
    rel = _rel(ds)
    ab = rf.poses_relativas_para_absolutas(rel)
    np.testing.assert_allclose(np.stack(ab), G[f"{ds}_absolute"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(np.stack(rf.poses_absolutas_para_relativas(ab)), G[f"{ds}_relative_back"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(rf.Calcular_Erro_LoopClosure(rel), G[f"{ds}_closure"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(rf.compor_duas_poses(rel[1], rel[0]), G[f"{ds}_compose01"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(rf.Transformar_de_volta(rel[0]), G[f"{ds}_inverse0"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("ds", DATASETS)
def test_lum_matches_reference_goldens(ds):
    rel = _rel(ds)
    n = len(rel)
    rots = [a[:3, :3] for a in G[f"{ds}_absolute"][1:]] + [G[f"{ds}_closure"][:3, :3]]
    Lb, tclos = rf.Montar_Vetor_Lb_translacoes(rel, rots)
    np.testing.assert_allclose(Lb, G[f"{ds}_Lb"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(tclos, G[f"{ds}_t_closure"], rtol=0, atol=1e-14)
    w = [1.0 + 0.1 * i for i in range(n)]
    np.testing.assert_array_equal(rf.Montar_Matriz_Diagonal_Pesos(w, n), G[f"{ds}_P"])
    np.testing.assert_allclose(np.stack(rf.reconstruir_Ts_para_origem_LUM(rel, [1.0] * n)), G[f"{ds}_lum"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(np.stack(rf.reconstruir_Ts_para_origem_LUM(rel, w)), G[f"{ds}_lum_weighted"], rtol=0, atol=1e-11)
    # the script-3 LUM is the same model with unit weights and an identity-first rotation list
    np.testing.assert_allclose(np.stack(rf.script3.reconstruir_Ts_para_origem_LUM(rel)), G[f"{ds}_lum"], rtol=0, atol=1e-11)
    dR, dt = rf.subtract_squared_poses(list(G[f"{ds}_absolute"]), list(G[f"{ds}_lum"]))
    np.testing.assert_allclose(dR, G[f"{ds}_dR"], atol=1e-15)
    np.testing.assert_allclose(dt, G[f"{ds}_dt"], atol=1e-15)
    assert pcr_amd.create_scales(4) == list(G["create_scales_4"])


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def test_quaternion_helpers():
    rng = np.random.default_rng(3)
    for _ in range(50):
        R = rf.rand_rotation_matrix(rng=rng)
        assert abs(np.linalg.det(R) - 1) < 1e-12
        q = rf.quat_from_rotation_matrix(R)
        assert abs(np.linalg.norm(q) - 1) < 1e-14 and q[0] >= 0
        np.testing.assert_allclose(rf.quat_as_rotation_matrix(q), R, atol=1e-13)
        np.testing.assert_allclose(rf.quat_as_rotation_matrix(3.0 * q), R, atol=1e-13)       # non-unit input
        R2 = rf.rand_rotation_matrix(rng=rng)
        q2 = rf.quat_from_rotation_matrix(R2)
        np.testing.assert_allclose(rf.quat_as_rotation_matrix(rf.quat_multiply(q, q2)), R @ R2, atol=1e-13)
        np.testing.assert_allclose(rf.quat_multiply(q, rf.quat_inverse(q)), [1, 0, 0, 0], atol=1e-14)
    # slerp: end points, half-way on a single-axis rotation, hemisphere independence
    qa = rf.quat_from_rotation_matrix(_rot([0, 0, 1], 0.2))
    qb = rf.quat_from_rotation_matrix(_rot([0, 0, 1], 1.0))
    np.testing.assert_allclose(rf.quat_as_rotation_matrix(rf.quat_slerp(qa, qb, 0, 1, 0.0)), _rot([0, 0, 1], 0.2), atol=1e-14)
    np.testing.assert_allclose(rf.quat_as_rotation_matrix(rf.quat_slerp(qa, qb, 0, 1, 1.0)), _rot([0, 0, 1], 1.0), atol=1e-14)
    np.testing.assert_allclose(rf.quat_as_rotation_matrix(rf.quat_slerp(qa, qb, 0, 1, 0.5)), _rot([0, 0, 1], 0.6), atol=1e-14)
    np.testing.assert_allclose(rf.quat_as_rotation_matrix(rf.quat_slerp(qa, -qb, 0, 1, 0.5)), _rot([0, 0, 1], 0.6), atol=1e-14)
    np.testing.assert_allclose(rf.quat_as_rotation_matrix(rf.quat_slerp(qa, qb, 2.0, 4.0, 3.0)), _rot([0, 0, 1], 0.6), atol=1e-14)
    T = rf.interpolar_duas_T(rf._pose(_rot([0, 0, 1], 0.2), [0, 0, 0]), rf._pose(_rot([0, 0, 1], 1.0), [2, 0, 4]), 0.25)
    np.testing.assert_allclose(T, rf._pose(_rot([0, 0, 1], 0.4), [0.5, 0, 1.0]), atol=1e-14)
    np.testing.assert_allclose(rf.transformar_quaternio_em_4x4(qa, [1, 2, 3]), rf._pose(_rot([0, 0, 1], 0.2), [1, 2, 3]), atol=1e-14)


def _closed_circuit(n, rng, rot_noise=0.0, trans_noise=0.0):
    """n relative poses of a loop that closes exactly (in the reference's composition rule), plus optional noise."""
    ab = [np.eye(4)]
    for _ in range(n - 1):
        ab.append(rf._pose(_rot(rng.normal(size=3), rng.uniform(0.05, 0.6)) @ ab[-1][:3, :3], ab[-1][:3, 3] + rng.normal(size=3)))
    rel = rf.poses_absolutas_para_relativas(ab + [np.eye(4)])            # n relative poses, the last one closes the loop
    if rot_noise or trans_noise:
        rel = [rf._pose(_rot(rng.normal(size=3), rot_noise) @ T[:3, :3], T[:3, 3] + trans_noise * rng.normal(size=3)) for T in rel]
    return ab, rel


@pytest.mark.parametrize("n", [3, 7, 40])
def test_exact_loop_is_a_fixed_point_of_every_refinement(n):
    rng = np.random.default_rng(n)
    ab, rel = _closed_circuit(n, rng)
    np.testing.assert_allclose(rf.Calcular_Erro_LoopClosure(rel), np.hstack((np.eye(3), np.zeros((3, 1)))), atol=1e-12)
    ones = [1.0] * n
    for poses in (rf.poses_relativas_para_absolutas(rel), rf.reconstruir_Ts_para_origem_LUM(rel, ones),
                  rf.reconstruir_Ts_para_origem_SLERP(rel), rf.reconstruir_Ts_para_origem_SLERP_LUM(rel, ones),
                  rf.script3.reconstruir_Ts_para_origem_LUM(rel), rf.script3.reconstruir_Ts_para_origem_SLERP(rel),
                  rf.script3.reconstruir_Ts_para_origem_SLERP_LUM(rel)):
        assert len(poses) == n
        np.testing.assert_allclose(np.stack(poses), np.stack(ab), atol=1e-11)


@pytest.mark.parametrize("n", [3, 8, 60])
def test_noisy_loop_properties(n):
    rng = np.random.default_rng(100 + n)
    ab, rel = _closed_circuit(n, rng, rot_noise=0.01, trans_noise=0.02)
    ones = [1.0] * n
    plain = rf.poses_relativas_para_absolutas(rel)
    # library and script-3 flavours are the same model
    for a, b in ((rf.reconstruir_Ts_para_origem_SLERP(rel), rf.script3.reconstruir_Ts_para_origem_SLERP(rel)),
                 (rf.reconstruir_Ts_para_origem_SLERP_LUM(rel, ones), rf.script3.reconstruir_Ts_para_origem_SLERP_LUM(rel)),
                 (rf.reconstruir_Ts_para_origem_LUM(rel, ones), rf.script3.reconstruir_Ts_para_origem_LUM(rel))):
        np.testing.assert_allclose(np.stack(a), np.stack(b), atol=1e-11)
    sl = rf.reconstruir_Ts_para_origem_SLERP_LUM(rel, ones)
    for T in sl:
        np.testing.assert_allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12)
        np.testing.assert_array_equal(T[3], [0, 0, 0, 1])
    # SLERP spreads the closure rotation: adjusted rotation k lies between the forward and the backward chain, a
    # fraction k/n of the way => its angular distance to the forward chain grows linearly with k
    closure = rf.Calcular_Erro_LoopClosure(rel)[:3, :3]
    total = np.arccos(np.clip((np.trace(closure) - 1) / 2, -1, 1))
    for k in range(1, n):
        d = sl[k][:3, :3] @ plain[k][:3, :3].T
        ang = np.arccos(np.clip((np.trace(d) - 1) / 2, -1, 1))
        assert abs(ang - total * k / n) < 1e-9
    # LUM spreads the translation misclosure uniformly: every residual block equals closure/n
    rots = [p[:3, :3] for p in plain[1:]] + [closure]
    Lb, tclos = rf.Montar_Vetor_Lb_translacoes(rel, rots)
    lum = rf.reconstruir_Ts_para_origem_LUM(rel, ones)
    X = np.concatenate([p[:3, 3] for p in lum[1:]]).reshape(-1, 1)
    V = (-rf._lum_design(n) @ X + Lb).reshape(n, 3)
    np.testing.assert_allclose(V, np.tile(tclos / n, (n, 1)), atol=1e-10)
    # a heavier weight on one observation shrinks its residual
    w = ones.copy(); w[1] = 100.0
    lumw = rf.reconstruir_Ts_para_origem_LUM(rel, w)
    Xw = np.concatenate([p[:3, 3] for p in lumw[1:]]).reshape(-1, 1)
    Vw = (-rf._lum_design(n) @ Xw + Lb).reshape(n, 3)
    assert np.linalg.norm(Vw[1]) < 0.05 * np.linalg.norm(V[1])


def test_three_cloud_hand_case():
    """n = 3, rotations about z only: everything has a closed form."""
    a, b, c = 0.30, 0.50, -0.77                                       # closure angle = a+b+c = 0.03
    rel = [rf._pose(_rot([0, 0, 1], a), [1, 0, 0]), rf._pose(_rot([0, 0, 1], b), [0, 1, 0]), rf._pose(_rot([0, 0, 1], c), [0, 0, 1])]
    e = a + b + c
    sl = rf.script3.reconstruir_Ts_para_origem_SLERP(rel)
    np.testing.assert_allclose(sl[1][:3, :3], _rot([0, 0, 1], a - e / 3), atol=1e-13)
    np.testing.assert_allclose(sl[2][:3, :3], _rot([0, 0, 1], a + b - 2 * e / 3), atol=1e-13)
    np.testing.assert_allclose(sl[1][:3, 3], [1, 0, 0], atol=1e-13)                      # t1 = I·t0
    np.testing.assert_allclose(sl[2][:3, 3], np.array([1, 0, 0]) + _rot([0, 0, 1], a - e / 3) @ [0, 1, 0], atol=1e-13)
    lum = rf.script3.reconstruir_Ts_para_origem_LUM(rel)
    l0, l1, l2 = np.array([1., 0, 0]), _rot([0, 0, 1], a) @ [0, 1, 0], _rot([0, 0, 1], a + b) @ [0, 0, 1]
    m = (l0 + l1 + l2) / 3
    np.testing.assert_allclose(lum[1][:3, 3], l0 - m, atol=1e-13)
    np.testing.assert_allclose(lum[2][:3, 3], l0 + l1 - 2 * m, atol=1e-13)
    dR, dt = rf.script3.subtract_squared_poses(lum, sl)
    dR0, _ = rf.subtract_squared_poses(lum, sl)
    np.testing.assert_allclose(np.array(dR) * np.sqrt(2), dR0)
    with pytest.raises(Exception):
        rf.subtract_squared_poses(lum, sl[:2])


def test_four_cloud_hand_case_slerp_lum():
    """n = 4, rotations about z only (they commute, so the reversed product order of the reference's composition does not show):
    SLERP + LUM has a closed form.  With relative rotations a, b, c, d whose sum misses a full turn by e, SLERP takes k/4 of e off the
    k-th absolute rotation, R_k = Rz(a + ... - k e / 4); LUM then spreads the translation misclosure of the observations
    l_k = R_k t_k uniformly: X_k = sum_(j<k) l_j - k mean(l).  Both flavours (script 3, ALL_FUNCTIONS with unit weights)."""
    a, b, c, d = 0.40, 1.10, 1.70, 3.05
    ts = [np.array([2.0, 0.0, 0.1]), np.array([0.0, 1.5, 0.0]), np.array([1.0, 1.0, 0.0]), np.array([0.3, 0.0, 2.0])]
    rel = [rf._pose(_rot([0, 0, 1], x), t) for x, t in zip((a, b, c, d), ts)]
    e = a + b + c + d - 2 * np.pi
    assert abs(e + 0.03318530717958623) < 1e-15
    cum = np.cumsum([0.0, a, b, c])
    R = [_rot([0, 0, 1], cum[k] - k * e / 4) for k in range(4)]
    obs = [R[k] @ ts[k] for k in range(4)]
    mean = sum(obs) / 4
    X = [sum(obs[:k], np.zeros(3)) - k * mean for k in range(4)]
    for out in (rf.script3.reconstruir_Ts_para_origem_SLERP_LUM(rel), rf.reconstruir_Ts_para_origem_SLERP_LUM(rel, [1.0] * 4)):
        assert len(out) == 4
        for k in range(4):
            np.testing.assert_allclose(out[k][:3, :3], R[k], atol=1e-14)
            np.testing.assert_allclose(out[k][:3, 3], X[k], atol=1e-14)
            np.testing.assert_array_equal(out[k][3], [0, 0, 0, 1])


def test_script3_and_library_variants_on_the_shipped_circuits():
    """Regression record (SURVEY f-2, nothing more can be pinned without numpy-quaternion / Open3D): on the shipped Facade and Courtyard
    relative poses the script-3 variants (3_Global...py:154-284) and the ALL_FUNCTIONS ones with unit weights (AF:538-667) are the SAME
    poses to 1e-14, and how far each adjustment moves the plain composition is recorded to 3 digits: LUM moves translations only,
    SLERP takes the closure angle (Facade 5.83e-3 rad, Courtyard 3.27e-3 rad) off the rotations, SLERP + LUM both."""
    import os
    from conftest import GOLDEN, pose_error
    R = pcr_amd.refinement
    record = {"facade": {"closure": (5.828e-3, 0.10133), "LUM": (0.0, 8.686e-2), "SLERP": (4.995e-3, 8.545e-3), "SLERP_LUM": (4.995e-3, 9.676e-2)},
              "courtyard": {"closure": (3.269e-3, 0.27531), "LUM": (0.0, 0.24089), "SLERP": (2.861e-3, 3.735e-2), "SLERP_LUM": (2.861e-3, 0.22699)}}
    for name, want in record.items():
        d = np.load(os.path.join(GOLDEN, f"poses_{name}.npz"))
        names = list(d["relative_names"]); n = len(names)
        rel = [d["relative"][names.index(f"pose_{i + 1}_{i}.txt")] for i in range(n - 1)] + [d["relative"][names.index(f"pose_0_{n - 1}.txt")]]
        plain = R.poses_relativas_para_absolutas(rel)
        ang, dt = pose_error(R.Calcular_Erro_LoopClosure(rel), np.eye(4))
        assert abs(ang - want["closure"][0]) < 2e-6 and abs(dt - want["closure"][1]) < 2e-5, (name, ang, dt)
        pairs = {"LUM": (R.reconstruir_Ts_para_origem_LUM(rel, np.ones(n)), R.script3.reconstruir_Ts_para_origem_LUM(rel)),
                 "SLERP": (R.reconstruir_Ts_para_origem_SLERP(rel), R.script3.reconstruir_Ts_para_origem_SLERP(rel)),
                 "SLERP_LUM": (R.reconstruir_Ts_para_origem_SLERP_LUM(rel, np.ones(n)), R.script3.reconstruir_Ts_para_origem_SLERP_LUM(rel))}
        for method, (lib, s3) in pairs.items():
            same = np.array([pose_error(x, y) for x, y in zip(lib, s3)]).max(0)
            assert same[0] < 1e-13 and same[1] < 1e-13, (name, method, same)
            moved = np.array([pose_error(x, y) for x, y in zip(s3, plain)]).max(0)
            assert abs(moved[0] - want[method][0]) < 2e-6 + 1e-3 * want[method][0] and abs(moved[1] - want[method][1]) < 2e-5 + 1e-3 * want[method][1], (name, method, moved)


def test_which_variant_made_the_shipped_absolute_poses():
    """VERDICT r1 #9 / SURVEY f-2: the reference ships `absolute_poses_FGR_GICP/{Facade,Courtyard}` next to the relative poses of
    the same circuits.  Every host-side variant (plain composition, LUM, SLERP, SLERP+LUM, in the ALL_FUNCTIONS and the script-3
    flavours) was run on the shipped relative poses: NONE reproduces the shipped absolute poses -- the closest stays 9e-3 rad /
    0.25 m away on Facade and 1.6e-2 rad / 0.95 m on Courtyard, and the relative poses implied by the shipped absolutes
    (A_i^-1 A_(i+1)) differ from the shipped relative poses by 8e-3 rad / 0.21 m and 2e-2 rad / 0.8 m.  The shipped absolute poses
    therefore come from another run (other relative poses, or the pose-graph optimiser, whose inputs are not shipped) and cannot pin
    these functions; the distances are recorded here so that the statement stays checked."""
    import os
    from conftest import GOLDEN, pose_error
    R = pcr_amd.refinement
    measured = {"facade": {"best": (9.08e-3, 0.248), "implied": (7.82e-3, 0.210)}, "courtyard": {"best": (1.645e-2, 0.953), "implied": (1.93e-2, 0.815)}}
    for name in ("facade", "courtyard"):
        d = np.load(os.path.join(GOLDEN, f"poses_{name}.npz"))
        names = list(d["relative_names"]); n = len(names)
        rel = [d["relative"][names.index(f"pose_{i + 1}_{i}.txt")] for i in range(n - 1)] + [d["relative"][names.index(f"pose_0_{n - 1}.txt")]]
        an = list(d["absolute_names"]); ab = [d["absolute"][an.index(f"pose{i}.txt")] for i in range(n)]
        assert np.allclose(ab[0], np.eye(4))
        variants = {"plain": R.poses_relativas_para_absolutas(rel), "LUM": R.reconstruir_Ts_para_origem_LUM(rel, np.ones(n)),
                    "SLERP": R.reconstruir_Ts_para_origem_SLERP(rel), "SLERP_LUM": R.reconstruir_Ts_para_origem_SLERP_LUM(rel, np.ones(n)),
                    "LUM_S3": R.script3.reconstruir_Ts_para_origem_LUM(rel), "SLERP_S3": R.script3.reconstruir_Ts_para_origem_SLERP(rel),
                    "SLERP_LUM_S3": R.script3.reconstruir_Ts_para_origem_SLERP_LUM(rel)}
        err = {k: np.array([pose_error(v[i], ab[i]) for i in range(n)]).max(0) for k, v in variants.items()}
        best_rot = min(e[0] for e in err.values()); best_tr = min(e[1] for e in err.values())
        assert best_rot > 5e-3 and best_tr > 0.2, (name, err)                                   # nobody reproduces them ...
        assert abs(best_rot - measured[name]["best"][0]) < 2e-4 and abs(best_tr - measured[name]["best"][1]) < 5e-3, (name, best_rot, best_tr)
        implied = np.array([pose_error(np.linalg.inv(ab[i]) @ ab[i + 1], rel[i]) for i in range(n - 1)]).max(0)
        assert abs(implied[0] - measured[name]["implied"][0]) < 3e-4 and abs(implied[1] - measured[name]["implied"][1]) < 5e-3, (name, implied)
        # ... although all of them stay in the neighbourhood (same circuit): a sanity bound on the host functions themselves
        assert max(e[0] for e in err.values()) < 3e-2 and max(e[1] for e in err.values()) < 1.5
