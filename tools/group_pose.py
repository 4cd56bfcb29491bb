"""Test helper: script-2 Multiscale_GICP (5 scales) on the golden NCLT pairs (different cloud sizes) from their shipped FGR poses,
through register_pairs_plan with lockstep groups of 1, 2, 3 and 8 pairs; one line per group size with the pose bits, iteration and
cloud counts and a digest of the correspondence sets.  With PCR_ICP_PPL fixed in the environment every line must be the same.
GROUP_POSE_RULE=af uses the radius_from_cloud_pair rule; GROUP_POSE_LOSS=l2 the smooth loss."""
import glob, hashlib, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
reg = P.registration
files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "nclt_pair_*.npz")))
gold = [np.load(f) for f in files]
work = [(P.PointCloud(g["source"]), P.PointCloud(g["target"]), g["T_fgr"]) for g in gold]
work = work + [(t, s, np.linalg.inv(T)) for s, t, T in work[:2]]        # 5 pairs when there are 3 golden ones: no group size divides it
vox = P.script2.create_scales(5)
loss = reg.L2Loss() if os.environ.get("GROUP_POSE_LOSS") == "l2" else reg.L1Loss()
rule = "af" if os.environ.get("GROUP_POSE_RULE") == "af" else "given"
sizes = [int(v) for v in os.environ.get("GROUP_POSE_SIZES", "1,2,3,8").split(",")]       # 0 = group=None (sized by the clouds, forms by the pair)
pair_forms = {"": None, "0": False, "1": True}[os.environ.get("GROUP_POSE_PAIR_FORMS", "")]
stage = os.environ.get("GROUP_POSE_STAGE", "gicp")          # "fgr+gicp": registro_FGR per pair (fixed seeds), then the group's GICP from its poses, information matrices
for g in sizes:
    rs = reg.register_pairs_plan(work, stage, vox, P.script2.max_correspondence_distances(vox), reg.TransformationEstimationForGeneralizedICP(loss),
                                 reg.ICPConvergenceCriteria(1e-6, 1e-6, 100), inflight=2, with_correspondences=True, group=(g if g > 0 else None), pair_forms=pair_forms, radius_rule=rule,
                                 fgr_seed=77, prior_from_fgr=(stage != "gicp"), info_max_dist=(0.1 if stage != "gicp" else 0.0))
    h = hashlib.sha256()
    for r in rs:
        h.update(np.asarray(r.transformation).tobytes()); h.update(np.ascontiguousarray(r.correspondence_set).tobytes())
        h.update(repr([(s["iterations"], s["n_clean"], s["max_dist"]) for s in r.scales]).encode())
        if stage != "gicp":
            h.update(np.asarray(r.fgr.transformation).tobytes()); h.update(np.asarray(r.information).tobytes())
    print(f"GROUP {h.hexdigest()} fitness " + " ".join(f"{r.fitness:.6f}" for r in rs))
    if g == 1 or g == 3:
        print(f"POSES{g} " + " ".join(repr(float(v)) for r in rs for v in np.asarray(r.transformation).reshape(16)))
