"""Golden vectors for the HOST-side pose algebra / LUM refinement.

Runs in the build container only: imports the reference's ALL_FUNCTIONS.py with its un-installed native dependencies
(open3d, quaternion, seaborn) replaced by EMPTY stub modules, calls only its pure-numpy functions on the shipped pose
files, and stores inputs + outputs.  No reference source is copied; the fixture is data.

    python tests/golden/make_golden_host.py [/root/reference]
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main(ref):
    for name in ("open3d", "quaternion", "seaborn"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, ref)
    import matplotlib
    matplotlib.use("Agg")
    with contextlib.redirect_stdout(io.StringIO()):
        import ALL_FUNCTIONS as AF
    out = {}
    for ds, n in (("Facade", 7), ("Courtyard", 8)):
        d = os.path.join(ref, "relative_poses_FGR_GICP", ds)
        files = sorted(os.listdir(d), key=lambda f: (int(f.split("_")[2].split(".")[0]), f))
        rel = [np.loadtxt(os.path.join(d, f)) for f in files]
        with contextlib.redirect_stdout(io.StringIO()):
            ab = AF.poses_relativas_para_absolutas(rel)
            back = AF.poses_absolutas_para_relativas(ab)
            closure = AF.Calcular_Erro_LoopClosure(rel)
            rots = [a[:3, :3] for a in ab[1:]] + [closure[:3, :3]]
            Lb, tclos = AF.Montar_Vetor_Lb_translacoes(rel, rots)
            P = AF.Montar_Matriz_Diagonal_Pesos([1.0 + 0.1 * i for i in range(len(rel))], len(rel))
            lum = AF.reconstruir_Ts_para_origem_LUM(rel, [1.0] * len(rel))
            lum_w = AF.reconstruir_Ts_para_origem_LUM(rel, [1.0 + 0.1 * i for i in range(len(rel))])
            dR, dt = AF.subtract_squared_poses(ab, lum)
        out[f"{ds}_names"] = np.array(files)
        out[f"{ds}_relative"] = np.stack(rel)
        out[f"{ds}_absolute"] = np.stack(ab)
        out[f"{ds}_relative_back"] = np.stack(back)
        out[f"{ds}_closure"] = closure
        out[f"{ds}_Lb"] = Lb
        out[f"{ds}_t_closure"] = tclos
        out[f"{ds}_P"] = P
        out[f"{ds}_lum"] = np.stack(lum)
        out[f"{ds}_lum_weighted"] = np.stack(lum_w)
        out[f"{ds}_dR"] = np.array(dR); out[f"{ds}_dt"] = np.array(dt)
        out[f"{ds}_compose01"] = AF.compor_duas_poses(rel[1], rel[0])
        out[f"{ds}_inverse0"] = AF.Transformar_de_volta(rel[0])
    out["create_scales_4"] = np.array(AF.create_scales(4))
    np.savez_compressed(os.path.join(HERE, "host_pose_algebra.npz"), **out)
    print("wrote host_pose_algebra.npz with", len(out), "arrays")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
