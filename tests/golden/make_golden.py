"""Generate tests/golden/*.npz from the reference's shipped DATA files (run in the build
container only; /root/reference does not exist on the GPU box).

Fixtures are data, not code: input clouds (float32 xyz from the binary PCDs), the FGR pose the
reference's stage 2 reads as its initial transform, and the multiscale-GICP pose its author
shipped for the same pair (SURVEY.md §8c "strong" pin).

    python tests/golden/make_golden.py [/root/reference]
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
pio = importlib.import_module("point-cloud-registration-with-global-refinement_amd.io")

GOLDEN_PAIRS = [0, 10, 25, 145, 465, 500, 865, 899]   # stable members of the pinned list; 0 = s1 -> s0, the pair BASELINE
                                                       # config 1 names; 899 = small and noisy (wide L1 attractor)


def main(ref):
    nclt = os.path.join(ref, "nuvens/nuvens_pre_processadas/NCLT")
    for i in GOLDEN_PAIRS:
        src = pio.read_pcd_xyz(os.path.join(nclt, f"s{i + 1}.pcd"))
        tgt = pio.read_pcd_xyz(os.path.join(nclt, f"s{i}.pcd"))
        T_fgr = pio.read_pose(os.path.join(ref, f"relative_poses_FGR/NCLT/pose_{i + 1}_{i}.txt"))
        T_gicp = pio.read_pose(os.path.join(ref, f"relative_poses_FGR_GICP/NCLT/pose_{i + 1}_{i}.txt"))
        np.savez_compressed(os.path.join(HERE, f"nclt_pair_{i:03d}.npz"), source=src, target=tgt, T_fgr=T_fgr,
                            T_gicp=T_gicp, pair=np.int64(i))
        print(f"pair {i}: source {src.shape[0]} pts, target {tgt.shape[0]} pts")
    # point counts of the 901 NCLT scans (PCD header field POINTS): the size distribution the cost-balanced sharding is tested on
    counts = []
    for i in range(901):
        with open(os.path.join(nclt, f"s{i}.pcd"), "rb") as f:
            head = f.read(512).decode("ascii", errors="replace")
        counts.append(int([l for l in head.splitlines() if l.startswith("POINTS")][0].split()[1]))
    np.save(os.path.join(HERE, "nclt_point_counts.npy"), np.asarray(counts, dtype=np.int32))
    print(f"NCLT point counts: min {min(counts)} max {max(counts)} total {sum(counts)}")
    # the whole Facade loop (terrestrial scanner, 7 clouds of 45k-84k points: BASELINE config 4's shipped data) with the FGR poses
    # stage 2 starts from and the GICP poses the author shipped.  Pair i registers cloud i+1 onto cloud i, the last one closes the
    # loop (cloud 0 onto cloud 6; file pose_0_6.txt).  The shipped Facade GICP poses were NOT made with script-2 parameters (the
    # oracle lands 1e-3 rad / 2 cm away on all 7 pairs), so they pin nothing by themselves: the loop is a HIP-vs-oracle parity case on
    # a second sensor and point density, and the input of the host-side global refinement (3_Global_Optimizations...py:292-358).
    fac = os.path.join(ref, "nuvens/nuvens_pre_processadas/Facade")
    names = [f"pose_{i + 1}_{i}.txt" for i in range(6)] + ["pose_0_6.txt"]
    loop = {f"s{i}": pio.read_pcd_xyz(os.path.join(fac, f"s{i}.pcd")) for i in range(7)}
    loop["T_fgr"] = np.stack([pio.read_pose(os.path.join(ref, "relative_poses_FGR/Facade", n)) for n in names])
    loop["T_gicp"] = np.stack([pio.read_pose(os.path.join(ref, "relative_poses_FGR_GICP/Facade", n)) for n in names])
    np.savez_compressed(os.path.join(HERE, "facade_loop.npz"), **loop)
    # pose chains for the host-side pose-algebra tests (tiny)
    for name, n in (("Facade", 7), ("Courtyard", 8)):
        rel = [pio.read_pose(os.path.join(ref, f"relative_poses_FGR_GICP/{name}/{f}"))
               for f in sorted(os.listdir(os.path.join(ref, f"relative_poses_FGR_GICP/{name}")))]
        names = sorted(os.listdir(os.path.join(ref, f"relative_poses_FGR_GICP/{name}")))
        ab_dir = os.path.join(ref, f"absolute_poses_FGR_GICP/{name}")
        ab_names = sorted(os.listdir(ab_dir))
        ab = [pio.read_pose(os.path.join(ab_dir, f)) for f in ab_names]
        np.savez_compressed(os.path.join(HERE, f"poses_{name.lower()}.npz"), relative=np.stack(rel),
                            relative_names=np.array(names), absolute=np.stack(ab), absolute_names=np.array(ab_names))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
