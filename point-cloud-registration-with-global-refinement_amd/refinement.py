"""Host-side global refinement of a closed circuit of relative poses.

This stage stays on the host in the reference too: it is a few hundred 4x4 products and one dense 3(n-1) solve.
Names, argument order and return shapes follow the reference; the numerics are restated, not copied:

* pose algebra .............................. ALL_FUNCTIONS.py:101-145, 826-833, 963-977
* closure error / rel->abs .................. ALL_FUNCTIONS.py:478-529  (S1:96-122, S2:46-72)
* LUM (translations, Lu & Milios) ........... ALL_FUNCTIONS.py:446-468, 597-629 (weighted) / 3_Global…py:133-146, 191-219 (unweighted)
* SLERP (rotations) ......................... ALL_FUNCTIONS.py:538-592 / 3_Global…py:154-185, 225-250
* SLERP+LUM ................................. ALL_FUNCTIONS.py:637-667 / 3_Global…py:258-284

Conventions (all preserved): `T_circuito[i]` maps cloud i+1 into cloud i and the LAST entry closes the loop
(cloud 0 -> cloud n-1); absolute rotation k is the product R[k-1]·…·R[1]·R[0] (the reference's order);
outputs are lists of n 4x4 float64 poses whose first element is the identity.

`numpy-quaternion` (un-vendored, absent here) is replaced by the small `Quaternion` helpers below with the same
semantics at the call sites the reference uses: Hamilton product, inverse, rotation-matrix conversion, and
`slerp(q1, q2, t1, t2, t_out)` which -- like that library -- first moves q2 into q1's hemisphere (chordal distance
> sqrt(2) => negate) and then evaluates (q2 q1^-1)^tau q1.

The module-level functions are the `ALL_FUNCTIONS` flavour (weights argument, closure returned); the `script3`
namespace holds the variants that stage 3 actually runs.
"""
import numpy as np

_AUX_0001 = np.array([0.0, 0.0, 0.0, 1.0])


# ------------------------------------------------------------------------------------------------ quaternions
def quat_from_rotation_matrix(R):
    """Unit quaternion (w, x, y, z) of a 3x3 rotation; Shepperd's branch on the largest diagonal term.  Sign is
    canonicalised to w >= 0 (the rotation, and every use the reference makes of the value, is sign-invariant)."""
    R = np.asarray(R, dtype=np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    d = np.array([R[0, 0], R[1, 1], R[2, 2], tr])
    k = int(np.argmax(d))
    if k == 3:
        q = np.array([1.0 + tr, R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    elif k == 0:
        q = np.array([R[2, 1] - R[1, 2], 1.0 + R[0, 0] - R[1, 1] - R[2, 2], R[0, 1] + R[1, 0], R[0, 2] + R[2, 0]])
    elif k == 1:
        q = np.array([R[0, 2] - R[2, 0], R[0, 1] + R[1, 0], 1.0 - R[0, 0] + R[1, 1] - R[2, 2], R[1, 2] + R[2, 1]])
    else:
        q = np.array([R[1, 0] - R[0, 1], R[0, 2] + R[2, 0], R[1, 2] + R[2, 1], 1.0 - R[0, 0] - R[1, 1] + R[2, 2]])
    q = q / np.linalg.norm(q)
    return -q if q[0] < 0 else q


def quat_as_rotation_matrix(q):
    """3x3 rotation of a (not necessarily unit) quaternion: R = I + 2/|q|^2 (w[v]x + [v]x^2)."""
    w, x, y, z = np.asarray(q, dtype=np.float64)
    s = 2.0 / (w * w + x * x + y * y + z * z)
    return np.array([
        [1.0 - s * (y * y + z * z), s * (x * y - w * z), s * (x * z + w * y)],
        [s * (x * y + w * z), 1.0 - s * (x * x + z * z), s * (y * z - w * x)],
        [s * (x * z - w * y), s * (y * z + w * x), 1.0 - s * (x * x + y * y)]])


def quat_multiply(a, b):
    """Hamilton product a*b; as rotations R(a*b) = R(a) @ R(b)."""
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def quat_inverse(q):
    q = np.asarray(q, dtype=np.float64)
    return np.array([q[0], -q[1], -q[2], -q[3]]) / float(q @ q)


def quat_power(q, tau):
    """q**tau through log/exp (principal branch); q is normalised first."""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q)
    v = np.linalg.norm(q[1:])
    if v < 1e-300:
        if q[0] >= 0:
            return np.array([1.0, 0.0, 0.0, 0.0])
        # q == -1: the library picks an arbitrary axis; take x
        return np.array([np.cos(np.pi * tau), np.sin(np.pi * tau), 0.0, 0.0])
    half = np.arctan2(v, q[0])
    return np.concatenate(([np.cos(half * tau)], np.sin(half * tau) * q[1:] / v))


def quat_slerp(q1, q2, t1, t2, t_out):
    """`quaternion_time_series.slerp(R1, R2, t1, t2, t_out)` for scalar t_out (AF:129, 558; S3:179-181)."""
    tau = (t_out - t1) / (t2 - t1)
    q1 = np.asarray(q1, dtype=np.float64)
    q2 = np.asarray(q2, dtype=np.float64)
    if np.linalg.norm(q1 - q2) > np.sqrt(2.0):
        q2 = -q2
    return quat_multiply(quat_power(quat_multiply(q2, quat_inverse(q1)), tau), q1)


# ------------------------------------------------------------------------------------------------ pose algebra
def _pose(R, t):
    T = np.empty((4, 4))
    T[:3, :3] = R
    T[:3, 3] = np.asarray(t, dtype=np.float64).reshape(3)
    T[3] = _AUX_0001
    return T


def transformar_quaternio_em_4x4(quaternio, translacao):
    """AF:101-105."""
    return _pose(quat_as_rotation_matrix(quaternio), translacao)


def Transformar_de_volta(T_4x4):
    """Inverse of a rigid pose (AF:109-113; `Invert_pose` S3:22-26)."""
    T_4x4 = np.asarray(T_4x4, dtype=np.float64)
    Rt = T_4x4[:3, :3].T
    return _pose(Rt, -Rt @ T_4x4[:3, 3])


Invert_pose = Transformar_de_volta


def interpolar_duas_T(T1, T2, t):
    """Pose between T1 (t=0) and T2 (t=1): linear translation, SLERP rotation (AF:118-134)."""
    T1 = np.asarray(T1, dtype=np.float64)
    T2 = np.asarray(T2, dtype=np.float64)
    q = quat_slerp(quat_from_rotation_matrix(T1[:3, :3]), quat_from_rotation_matrix(T2[:3, :3]), 0, 1, t)
    return _pose(quat_as_rotation_matrix(q), T1[:3, 3] * (1 - t) + T2[:3, 3] * t)


def compor_duas_poses(T21, T10):
    """AF:142-147: R20 = R21 @ R10, t20 = R10 @ t21 + t10 (the reference's composition rule, kept as is)."""
    T21 = np.asarray(T21, dtype=np.float64)
    T10 = np.asarray(T10, dtype=np.float64)
    return _pose(T21[:3, :3] @ T10[:3, :3], T10[:3, :3] @ T21[:3, 3] + T10[:3, 3])


def _rotacoes_origem(T_circuito):
    """[R0, R1·R0, …, R(n-1)·…·R0] (AF:480-486 / 506-512 / 600-606).  Each product is formed as ((I·R[k])·R[k-1])·…·R[0],
    the reference's association, so the result is bit-comparable with the stub-import goldens."""
    out = []
    for k in range(len(T_circuito)):
        acc = np.identity(3)
        for j in range(k, -1, -1):
            acc = acc @ np.asarray(T_circuito[j], dtype=np.float64)[:3, :3]
        out.append(acc)
    return out


def Montar_Matriz_Diagonal_Pesos(Pesos, n_nuvens):
    """AF:446-452: diag(w0,w0,w0,w1,w1,w1,…)."""
    return np.diagflat(np.repeat(np.asarray(Pesos[:n_nuvens], dtype=np.float64), 3))


def Montar_Vetor_Lb_translacoes(T_circuito, lista_rotacoes_origem):
    """AF:459-468: observations l0 = t0, l(i+1) = R_abs[i]·t(i+1); returns [Lb (3n x 1), closure translation]."""
    blocks = [np.asarray(T_circuito[0], dtype=np.float64)[:3, 3]]
    closure = blocks[0].copy()
    for i in range(len(T_circuito) - 1):
        b = lista_rotacoes_origem[i] @ np.asarray(T_circuito[i + 1], dtype=np.float64)[:3, 3]
        closure = closure + b
        blocks.append(b)
    return [np.concatenate(blocks).reshape(-1, 1), closure]


def Calcular_Erro_LoopClosure(T_circuito, verbose=False):
    """Composition of the whole circuit as a 3x4 [R | t]; ideally [I | 0] (AF:478-499).  The reference prints the
    pose and its Frobenius distance to the identity; here only with verbose=True."""
    rots = _rotacoes_origem(T_circuito)
    _, t = Montar_Vetor_Lb_translacoes(T_circuito, rots)
    closure = np.hstack((rots[-1], t.reshape(3, 1)))
    if verbose:
        print(f"POSE Closure error:\n{closure}")
        print(f"Distancia (Frobenious) para a identidade:\n{np.linalg.norm(rots[-1] - np.identity(3), 'fro')}")
    return closure


def poses_relativas_para_absolutas(T_circuito):
    """AF:503-529: plain composition; identity first, the closure pose dropped."""
    rots = _rotacoes_origem(T_circuito)
    t = np.asarray(T_circuito[0], dtype=np.float64)[:3, 3]
    poses = [np.identity(4), _pose(rots[0], t)]
    for i in range(len(T_circuito) - 1):
        t = rots[i] @ np.asarray(T_circuito[i + 1], dtype=np.float64)[:3, 3] + t
        poses.append(_pose(rots[i + 1], t))
    del poses[-1]
    return poses


def poses_absolutas_para_relativas(poses_absolutas):
    """AF:826-833: n absolute poses -> n-1 relative ones (no closing pose)."""
    n = len(poses_absolutas)
    inv = [Transformar_de_volta(poses_absolutas[i]) for i in range(n - 1)]
    return [compor_duas_poses(poses_absolutas[i + 1], inv[i]) for i in range(n - 1)]


def subtract_squared_poses(list_poses_1, list_poses_2):
    """Per-pose Frobenius distance of rotations and Euclidean distance of translations (AF:963-977).  The script-3
    copy divides d_R by sqrt(2) (S3:61-77) -- see `script3.subtract_squared_poses`."""
    if len(list_poses_1) != len(list_poses_2):
        raise Exception("The list of poses should be the same size")
    dR, dt = [], []
    for a, b in zip(list_poses_1, list_poses_2):
        d = (np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2
        dR.append(float(np.sqrt(d[:3, :3].sum())))
        dt.append(float(np.sqrt(d[:3, 3].sum())))
    return dR, dt


def rand_rotation_matrix(deflection=1.0, rng=None):
    """Arvo's uniform random rotation (AF:941-957); draws three uniforms from `rng` (numpy global state if None, as
    the reference)."""
    theta, phi, z = (np.random.uniform(size=(3,)) if rng is None else rng.uniform(size=(3,)))
    theta = theta * 2.0 * deflection * np.pi
    phi = phi * 2.0 * np.pi
    z = z * 2.0 * deflection
    r = np.sqrt(z)
    V = np.array([np.sin(phi) * r, np.cos(phi) * r, np.sqrt(2.0 - z)])
    st, ct = np.sin(theta), np.cos(theta)
    R = np.array(((ct, st, 0.0), (-st, ct, 0.0), (0.0, 0.0, 1.0)))
    return (np.outer(V, V) - np.eye(3)).dot(R)


# ------------------------------------------------------------------------------------------------ LUM
def _lum_design(n):
    """A (3n x 3(n-1)): +I on the block diagonal, -I on the first block sub-diagonal (AF:611-613, S3:204-206):
    l0 = x1, lk = x(k+1) - xk, l(n-1) = -x(n-1)."""
    A = np.zeros((3 * n, 3 * (n - 1)))
    idx = np.arange(3 * (n - 1))
    A[idx, idx] = 1.0
    A[idx + 3, idx] = -1.0
    return A


def _lum_solve(Lb, n, P=None):
    A = _lum_design(n)
    if P is None:
        N = np.linalg.inv(A.T @ A)
        X = N @ (A.T @ Lb)
    else:
        N = np.linalg.inv(A.T @ P @ A)
        X = N @ (A.T @ P @ Lb)
    V = -A @ X + Lb
    return X, V


def reconstruir_Ts_para_origem_LUM(T_circuito, Pesos, verbose=False):
    """Translations adjusted by weighted least squares, rotations composed unadjusted (AF:597-629)."""
    n = len(T_circuito)
    rots = _rotacoes_origem(T_circuito)
    Lb, _ = Montar_Vetor_Lb_translacoes(T_circuito, rots)
    P = Montar_Matriz_Diagonal_Pesos(Pesos, n)
    X, V = _lum_solve(Lb, n, P)
    if verbose:
        print(f"Sigma_Posterior = Vt*P*V of LUM = {(V.T @ P @ V) / 3} ")
    return [np.identity(4)] + [_pose(rots[i], X[3 * i:3 * i + 3]) for i in range(n - 1)]


# ------------------------------------------------------------------------------------------------ SLERP
def Ajustamento_Quaternios_SLERP(lista_quat):
    """AF:538-560.  Forward absolute rotations a_i = q_i…q_0; the same rotations reached the other way round the loop
    b_i = a_i · closure^-1; adjusted rotation i = slerp(a_i, b_i, (i+1)/n).  Returns [n-1 quaternions, closure]."""
    n = len(lista_quat)
    origem = []
    for k in range(n):
        r = np.array([1.0, 0.0, 0.0, 0.0])
        for j in range(k, -1, -1):
            r = quat_multiply(r, lista_quat[j])
        origem.append(r)
    closure = origem[n - 1]
    cinv = quat_inverse(closure)
    ajustado = [quat_slerp(origem[i], quat_multiply(origem[i], cinv), 0, 1, (i + 1) / n) for i in range(n - 1)]
    return [ajustado, closure]


def _quats_of(T_circuito):
    return [quat_from_rotation_matrix(np.asarray(T, dtype=np.float64)[:3, :3]) for T in T_circuito]


def reconstruir_Ts_para_origem_SLERP(T_circuito):
    """Rotations adjusted by SLERP, translations re-accumulated with the adjusted rotations (AF:569-592)."""
    n = len(T_circuito)
    ajustado, _ = Ajustamento_Quaternios_SLERP(_quats_of(T_circuito))
    Rs = [quat_as_rotation_matrix(q) for q in ajustado]
    t = np.asarray(T_circuito[0], dtype=np.float64)[:3, 3]
    ts = [t]
    for i in range(n - 1):
        t = Rs[i] @ np.asarray(T_circuito[i + 1], dtype=np.float64)[:3, 3] + t
        ts.append(t)
    return [np.identity(4)] + [_pose(Rs[i], ts[i]) for i in range(n - 1)]


def reconstruir_Ts_para_origem_SLERP_LUM(T_circuito, Pesos, verbose=False):
    """SLERP-adjusted rotations feeding the weighted LUM translation solve (AF:637-667)."""
    n = len(T_circuito)
    ajustado, _ = Ajustamento_Quaternios_SLERP(_quats_of(T_circuito))
    Rs = [quat_as_rotation_matrix(q) for q in ajustado]
    Lb, _ = Montar_Vetor_Lb_translacoes(T_circuito, Rs)
    P = Montar_Matriz_Diagonal_Pesos(Pesos, n)
    X, V = _lum_solve(Lb, n, P)
    if verbose:
        print(f"Sigma_Posterior = Vt*P*V of LUM = {V.T @ P @ V / 3} GL = 3")
    return [np.identity(4)] + [_pose(Rs[i], X[3 * i:3 * i + 3]) for i in range(n - 1)]


# ------------------------------------------------------------------------------------------------ script-3 variants
class script3:
    """The private copies `3_Global_Optimizations_in_NCLT_dataset.py` runs (unweighted; identity-first lists)."""

    Invert_pose = staticmethod(Transformar_de_volta)

    @staticmethod
    def subtract_squared_poses(list_poses_1, list_poses_2):
        """S3:61-77: rotation distance scaled by 1/sqrt(2)."""
        dR, dt = subtract_squared_poses(list_poses_1, list_poses_2)
        return [d / np.sqrt(2.0) for d in dR], dt

    @staticmethod
    def Montar_Vetor_Lb_translacoes(T_circuito, lista_rotacoes_origem):
        """S3:133-146: Lb block i = R_abs[i]·t_i with R_abs[0] = I; returns the (3n x 1) column only."""
        return np.concatenate([lista_rotacoes_origem[i] @ np.asarray(T_circuito[i], dtype=np.float64)[:3, 3]
                               for i in range(len(T_circuito))]).reshape(-1, 1)

    @staticmethod
    def Ajustamento_Quaternios_SLERP(relative_quat):
        """S3:154-185: forward a_i = q_(i-1)·a_(i-1); backward products of the tail, inverted; adjusted rotation i =
        slerp(a_i, (q_(n-1)…q_i)^-1, i/n); identity prepended (n quaternions out)."""
        n = len(relative_quat)
        fwd, bwd_inv = [], []
        a = np.array([1.0, 0.0, 0.0, 0.0])
        b = np.array([1.0, 0.0, 0.0, 0.0])
        for i in range(1, n):
            a = quat_multiply(relative_quat[i - 1], a)
            b = quat_multiply(b, relative_quat[-i])
            fwd.append(a)
            bwd_inv.append(quat_inverse(b))
        out = [np.array([1.0, 0.0, 0.0, 0.0])]
        for i in range(1, n):
            out.append(quat_slerp(fwd[i - 1], bwd_inv[-i], 0, 1, i / n))
        return out

    @staticmethod
    def _absolute_R(T_circuito):
        out, acc = [], np.identity(3)
        for T in T_circuito:
            out.append(acc)
            acc = np.asarray(T, dtype=np.float64)[:3, :3] @ acc
        return out

    @staticmethod
    def reconstruir_Ts_para_origem_LUM(T_circuito):
        """S3:191-219."""
        n = len(T_circuito)
        Rs = script3._absolute_R(T_circuito)
        X, _ = _lum_solve(script3.Montar_Vetor_Lb_translacoes(T_circuito, Rs), n)
        return [np.identity(4)] + [_pose(Rs[i], X[3 * (i - 1):3 * i]) for i in range(1, n)]

    @staticmethod
    def reconstruir_Ts_para_origem_SLERP(T_circuito):
        """S3:225-250."""
        qs = script3.Ajustamento_Quaternios_SLERP(_quats_of(T_circuito))
        poses, t = [], np.zeros(3)
        for i in range(len(T_circuito)):
            R = quat_as_rotation_matrix(qs[i])
            poses.append(_pose(R, t))
            t = R @ np.asarray(T_circuito[i], dtype=np.float64)[:3, 3] + t
        return poses

    @staticmethod
    def reconstruir_Ts_para_origem_SLERP_LUM(T_circuito):
        """S3:258-284."""
        n = len(T_circuito)
        Rs = [quat_as_rotation_matrix(q) for q in script3.Ajustamento_Quaternios_SLERP(_quats_of(T_circuito))]
        X, _ = _lum_solve(script3.Montar_Vetor_Lb_translacoes(T_circuito, Rs), n)
        return [np.identity(4)] + [_pose(Rs[i], X[3 * (i - 1):3 * i]) for i in range(1, n)]
