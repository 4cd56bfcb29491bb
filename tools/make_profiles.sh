#!/bin/bash
# Produces the round's committed profile evidence (run on the GPU box through gpurun):
#   profiles/rNN_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the DEFAULT bench command
#   profiles/rNN_pmc_hbm_traffic.csv      per-kernel HBM traffic from separate --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC hit/miss)
#   profiles/rNN_traffic.json             the same numbers keyed by kernel (bench.py reads it for roofline.traffic)
#   profiles/rNN_config5_kernel_stats.csv / rNN_nclt_kernel_stats.csv: the same summary for config 5 (2M points, 5 scales) and NCLT-size pairs
# usage: tools/make_profiles.sh r05
TAG=${1:-r05}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/profiles_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/profiles" "$ROOT/gpurun_out/profiles_export"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 2 --warmup 1 > "$OUT/bench_stats.log" 2> "$OUT/bench_stats.err" || exit 1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$ROOT/profiles/${TAG}_bench_kernel_stats.csv"
python3 "$ROOT/tools/icp_gap_hist.py" "$OUT/stats" "$ROOT/profiles/${TAG}_icp_gaps_inflight.txt" > /dev/null
python3 "$ROOT/tools/trace_overview.py" "$OUT/stats" 0.4 > "$ROOT/profiles/${TAG}_trace_overview_inflight.txt"
python3 "$ROOT/tools/trace_bins.py" "$OUT/stats" > "$ROOT/profiles/${TAG}_timeline_bins.txt"
# config 2's FGR variant (registro_FGR + the same GICP): kernel summary of the same command the bench line of that variant comes from
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fgr" -o run -- python3 "$ROOT/bench.py" --variant fgr --no-cpu-baseline --steps 1 --warmup 1 --pairs-per-step 16 > "$OUT/bench_fgr.log" 2> "$OUT/bench_fgr.err" || exit 1
cp "$(find "$OUT/fgr" -name '*kernel_stats.csv' | head -1)" "$ROOT/profiles/${TAG}_fgr_kernel_stats.csv"
grep '^{"metric' "$OUT/bench_fgr.log" | tail -1 > "$ROOT/profiles/${TAG}_bench_line_fgr_under_rocprof.json"
# the script-1 FGR stage on the shipped-size golden NCLT clouds in lockstep FGR groups (24 pairs per group, 4 groups in flight: the default rule)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fgrn" -o run -- python3 "$ROOT/tools/fgr_group_sweep.py" 24x4 > "$OUT/fgr_nclt.log" 2> "$OUT/fgr_nclt.err" || exit 1
cp "$(find "$OUT/fgrn" -name '*kernel_stats.csv' | head -1)" "$ROOT/profiles/${TAG}_fgr_nclt_groups_kernel_stats.csv"
grep '^fgr_group' "$OUT/fgr_nclt.log" | tail -1 > "$ROOT/profiles/${TAG}_fgr_nclt_groups_line.txt"
# the script-2 five-scale GICP stage on the same scans (24 x 4)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gicpn" -o run -- python3 "$ROOT/tools/gicp_nclt_sweep.py" 24x4 > "$OUT/gicp_nclt.log" 2> "$OUT/gicp_nclt.err" || exit 1
cp "$(find "$OUT/gicpn" -name '*kernel_stats.csv' | head -1)" "$ROOT/profiles/${TAG}_gicp_nclt_groups_kernel_stats.csv"
grep '^gicp group' "$OUT/gicp_nclt.log" | tail -1 > "$ROOT/profiles/${TAG}_gicp_nclt_groups_line.txt"
python3 "$ROOT/tools/trace_bins.py" "$OUT/gicpn" > "$ROOT/profiles/${TAG}_gicp_nclt_timeline_bins.txt"
# one pair at a time: gaps of the iteration chain without other pairs
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/solo" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 1 --warmup 1 --pairs-per-step 8 --inflight 1 --group 1 > "$OUT/bench_solo.log" 2> "$OUT/bench_solo.err" || exit 1
python3 "$ROOT/tools/icp_gap_hist.py" "$OUT/solo" "$ROOT/profiles/${TAG}_icp_gaps_solo.txt" > /dev/null
cp "$(find "$OUT/solo" -name '*kernel_stats.csv' | head -1)" "$ROOT/profiles/${TAG}_solo_kernel_stats.csv"
# config 5 (2M-point clouds, 5 scales, 64-NN normals) and NCLT-size pairs (20k points, the default group rule): kernel summaries of the commands
# whose bench lines are stored next to them
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5" -o run -- python3 "$ROOT/bench.py" --config5 --pairs-per-step 8 --base-pairs 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_c5.log" 2> "$OUT/bench_c5.err" || exit 1
cp "$(find "$OUT/c5" -name '*kernel_stats.csv' | head -1)" "$ROOT/profiles/${TAG}_config5_kernel_stats.csv"
grep '^{"metric' "$OUT/bench_c5.log" | tail -1 > "$ROOT/profiles/${TAG}_bench_line_config5_under_rocprof.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/nclt" -o run -- python3 "$ROOT/bench.py" --points 20000 --pairs-per-step 192 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_nclt.log" 2> "$OUT/bench_nclt.err" || exit 1
cp "$(find "$OUT/nclt" -name '*kernel_stats.csv' | head -1)" "$ROOT/profiles/${TAG}_nclt_kernel_stats.csv"
grep '^{"metric' "$OUT/bench_nclt.log" | tail -1 > "$ROOT/profiles/${TAG}_bench_line_nclt_under_rocprof.json"
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$OUT/pmc_$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$d" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 1 --warmup 1 --pairs-per-step 4 --inflight 1 --group 1 > "$d.log" 2> "$d.err" || exit 1
  echo "pmc pass [$pass] done"
done
# the same three passes on the DEFAULT path (lockstep groups of six: k_knn_wave_batchp, k_icp_fused_b<1024>), 12 pairs per step
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$OUT/pmcg_$(echo $pass | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$d" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 1 --warmup 1 --pairs-per-step 12 --inflight 2 > "$d.log" 2> "$d.err" || exit 1
  echo "pmc pass (groups) [$pass] done"
done
python3 - "$OUT" "$ROOT/profiles/${TAG}" <<'PY'
import csv, glob, json, sys, collections
out, dst = sys.argv[1], sys.argv[2]
def norm(name):                      # whatever the tile, the search form and the argument form
    if "k_icp_fused" in name: return "k_icp_fused"
    if name.startswith("void k_icp_nn<") or name.startswith("k_icp_nn("): return "k_icp_nn"
    if "k_icp_nn_g" in name: return "k_icp_nn_g"
    if "k_icp_lin_g" in name: return "k_icp_lin_g"
    if "k_normals_from_lists_batchp" in name: return "k_normals_from_lists_batchp"
    if name.startswith("void k_knn_wave_batchp<0, 30>"): return "k_knn_wave_batchp<SOR,30>"
    if name.startswith("void k_knn_list_batchp<0, 4>"): return "k_knn_list_batchp<SOR,4>"
    if name.startswith("void k_knn_batchp<1, 4>"): return "k_knn_batchp<NORMALS,4>"
    if name.startswith("void k_knn_list_batchp<1, 4>"): return "k_knn_list_batchp<NORMALS,4>"
    return {"void k_icp_iter<0>(IcpArgs)": "k_icp_iter<GICP>", "void k_knn_batch<0, 4>(KnnBatch)": "k_knn_batch<SOR,4>", "void k_knn_batch<1, 4>(KnnBatch)": "k_knn_batch<NORMALS,4>",
            "k_normals_from_lists_batch(NflBatch)": "k_normals_from_lists_batch", "k_grid_build(GridBuildDesc const*)": "k_grid_build"}.get(name, name.split("(")[0])
def load(pat):
    f = glob.glob(f"{out}/{pat}/**/*counter_collection.csv", recursive=True)[0]
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = norm(r["Kernel_Name"])
        d[name][r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
    return d
rows, js = [], {}
for group_form, names in ((False, ["k_icp_fused", "k_icp_nn", "k_icp_iter<GICP>", "k_knn_batch<SOR,4>", "k_knn_batch<NORMALS,4>", "k_normals_from_lists_batch", "k_grid_build", "k_rs_scatter"]),
                          (True, ["k_icp_fused", "k_knn_wave_batchp<SOR,30>", "k_knn_list_batchp<SOR,4>", "k_knn_list_batchp<NORMALS,4>", "k_normals_from_lists_batchp", "k_icp_nn_g", "k_icp_lin_g"])):
  pre = "pmcg_" if group_form else "pmc_"
  fetch, write, tcc = load(pre + "FETCH_SIZE"), load(pre + "WRITE_SIZE"), load(pre + "TCC_HIT_sum_TCC_MISS_sum")
  for name in names:
      full = name
      live = lambda lst: [v for v, dur in lst if dur > 6.0]            # launches after 'done' return at once: not live
      f = live(fetch[full]["FETCH_SIZE"]); w = live(write[full]["WRITE_SIZE"])
      h = live(tcc[full]["TCC_HIT_sum"]); m = live(tcc[full]["TCC_MISS_sum"])
      us = [dur for v, dur in fetch[full]["FETCH_SIZE"] if dur > 6.0]
      if not f or not w or not h: continue
      fk, wk = sum(f) / len(f), sum(w) / len(w)
      hbm = int((2.0 * fk + wk) * 1024)                                # gfx950: FETCH_SIZE counts 64 B per 128-B request (guide, HBM section)
      hit = sum(h) / max(1.0, sum(h) + sum(m))
      label = name + (" [default path: lockstep group of 6]" if group_form else " [one pair at a time]")
      rows.append([label, len(f), round(fk, 1), round(wk, 1), hbm, round(hit, 3), round(sum(us) / len(us), 1)])
      js[("groups/" if group_form else "") + name] = {"hbm_bytes_per_launch": hbm, "fetch_kb": fk, "write_kb": wk, "l2_hit": hit}
with open(dst + "_pmc_hbm_traffic.csv", "w", newline="") as fcsv:
    wr = csv.writer(fcsv); wr.writerow(["kernel", "live_launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "hbm_bytes_per_launch_corrected", "L2_hit_rate", "avg_us_live_under_pmc"]); wr.writerows(rows)
json.dump(js, open(dst + "_traffic.json", "w"), indent=1)
for r in rows: print(r)
PY
# VALU issue per kernel and pair (one pair at a time, kernels serialised by the counter collection): which kernels the chip's VALU time goes to
d=$OUT/pmc_valu
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d "$d" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 1 --warmup 1 --pairs-per-step 4 --inflight 1 --group 1 > "$d.log" 2> "$d.err" || exit 1
python3 - "$OUT" "$ROOT/profiles/${TAG}" <<'PY'
import csv, glob, sys, collections
out, dst = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/pmc_valu/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"]][r["Counter_Name"]].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
npairs = max(1.0, len(d["void k_knn_batch<0, 4>(KnnBatch)"]["SQ_INSTS_VALU"]) / 2)          # two batched SOR searches per pair
rows = sorted(((sum(x[0] for x in v["SQ_INSTS_VALU"]), k, len(v["SQ_INSTS_VALU"]), sum(x[1] for x in v["SQ_INSTS_VALU"])) for k, v in d.items()), reverse=True)
tot = sum(r[0] for r in rows)
PEAK = 256 * 4 * 2400 / 4.0                                                                  # wave64 VALU instructions per us: 1024 SIMDs, 4 cycles each at 2.4 GHz -- the MEASURED rate of
                                                                                             # three-source VOP3 (v_med3 / v_fma, profiles/r04_valu_rate.txt); two-source VOP2 issue at ~2.4 cycles (1024 k/us)
with open(dst + "_pmc_valu_per_pair.csv", "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "launches_per_pair", "kernel_ms_per_pair_serialised", "valu_wave_instructions_per_pair_M", "share_of_valu_instructions", "valu_instructions_per_us", "share_of_peak_issue_614k_per_us"])
    for insts, k, n, dur in rows[:20]:
        w.writerow([k[:60], round(n / npairs, 1), round(dur / 1e3 / npairs, 3), round(insts / 1e6 / npairs, 1), round(insts / tot, 3), int(insts / dur), round(insts / dur / PEAK, 3)])
    w.writerow(["TOTAL", "", round(sum(r[3] for r in rows) / 1e3 / npairs, 3), round(tot / 1e6 / npairs, 1), 1.0, "", ""])
PY
# the same table for the DEFAULT bench (lockstep groups: wavefront k-NN kernel, by-value fused kernel) -- what the headline number runs
d=$OUT/pmc_valu_groups
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d "$d" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 1 --warmup 1 > "$d.log" 2> "$d.err" || exit 1
python3 - "$OUT" "$ROOT/profiles/${TAG}" <<'PY'
import csv, glob, sys, collections, json
out, dst = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/pmc_valu_groups/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(lambda: [0.0, 0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "SQ_INSTS_VALU": continue
    e = d[r["Kernel_Name"].split("(")[0]]; e[0] += float(r["Counter_Value"]); e[1] += 1; e[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
line = json.loads([l for l in open(f"{out}/pmc_valu_groups.log") if l.startswith('{"metric')][-1])
pps = line["config"]["pairs_per_step"]
npairs = float(pps * (line["steps"] + max(line["warmup"], 1)) + 5)          # timed + warm-up steps + the single-pair steps of the roofline measurement
tot = sum(e[0] for e in d.values())
PEAK = 256 * 4 * 2400 / 4.0
with open(dst + "_pmc_valu_per_pair_groups.csv", "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "launches", "valu_wave_instructions_per_pair_M", "share_of_valu_instructions", "valu_instructions_per_us_serialised", "share_of_peak_issue_614k_per_us"])
    for k, e in sorted(d.items(), key=lambda x: -x[1][0])[:20]:
        w.writerow([k[:60], e[1], round(e[0] / 1e6 / npairs, 1), round(e[0] / tot, 3), int(e[0] / e[2]), round(e[0] / e[2] / PEAK, 3)])
    w.writerow(["TOTAL", "pairs counted: %d" % npairs, round(tot / 1e6 / npairs, 1), 1.0, "", ""])
PY
grep '^{"metric' "$OUT/bench_stats.log" | tail -1 > "$ROOT/profiles/${TAG}_bench_line_under_rocprof.json"
# gpurun only merges gpurun_out/ back: export the files to commit there as well
cp "$ROOT"/profiles/${TAG}_* "$ROOT/gpurun_out/profiles_export/"
head -12 "$ROOT/profiles/${TAG}_bench_kernel_stats.csv" | cut -c1-150
# the raw traces are large (gpurun merges at most 64 MiB back): keep the summaries only
find "$OUT" \( -name '*kernel_trace.csv' -o -name '*counter_collection.csv' -o -name '*.db' \) -delete
du -sh "$OUT"
