"""Copy the outputs of tools/refresh_profiles.sh (gpurun_out/refresh) into profiles/ and write profiles/<tag>_SUMMARY.md.
Usage: python tools/make_summary.py [tag]   (tag default r01)"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = os.path.join(ROOT, "profiles")


def cp(src, dst):
    shutil.copyfile(os.path.join(SRC, src), os.path.join(P, dst))


cp("bench_full.json", tag + "_bench_full.json")
cp("bench_hotpath_only.json", tag + "_bench_hotpath_only.json")
cp("bench_s2.json", tag + "_bench_s2.json")
if os.path.exists(os.path.join(SRC, "bench_hotpath_graph.json")):
    cp("bench_hotpath_graph.json", tag + "_bench_hotpath_graph.json")
for extra in ("bench_full_graph.json", "bench_s2_cfg4.json", "bench_ddp1.json", "eval_bench.json", "bench_full_two_renders.json",
              "bench_s2_two_renders.json", "step_kernels.jsonl"):
    if os.path.exists(os.path.join(SRC, extra)) and os.path.getsize(os.path.join(SRC, extra)) > 0:
        cp(extra, tag + "_" + extra)
if os.path.exists(os.path.join(SRC, "valu_ubench.log")):
    cp("valu_ubench.log", tag + "_valu_ubench.log")
cp("stats/t_kernel_stats.csv", tag + "_bench_kernel_stats.csv")
cp("traffic/traffic.json", tag + "_traffic.json")
cp("traffic/traffic.json", "traffic.json")
if os.path.exists(os.path.join(SRC, "pmc", "pmc_summary.json")):      # tools/r4/pmc_passes.py: SQ / SQC / TCC / TCP counters per raster kernel
    cp("pmc/pmc_summary.json", tag + "_pmc_step_kernels.json")
with open(os.path.join(P, tag + "_microbench.log"), "w") as f:
    for name in ("microbench.log", "kernel_only.log"):
        if os.path.exists(os.path.join(SRC, name)):
            f.write("".join(l for l in open(os.path.join(SRC, name)) if "amdgpu.ids" not in l))

full = json.load(open(os.path.join(SRC, "bench_full.json")))
hot = json.load(open(os.path.join(SRC, "bench_hotpath_only.json")))
s2 = json.load(open(os.path.join(SRC, "bench_s2.json")))
rows = list(csv.DictReader(open(os.path.join(SRC, "stats", "t_kernel_stats.csv"))))
steps = 45.0
total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
ours = [r for r in rows if "(anonymous namespace)::k_" in r["Name"]]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:92]
tr = json.load(open(os.path.join(SRC, "traffic", "traffic.json")))
L = []
L.append("# Round %s profile summary (1x MI355X)\n" % tag.lstrip("r0"))
L.append("All files in this directory are produced on the GPU box by `tools/refresh_profiles.sh` and copied here by "
         "`tools/make_summary.py`.\n")
L.append("Command behind the kernel table: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py "
         "--cpu-baseline 0` (10 warm-up + 30 timed + 5 profile-pass steps of the full train_s1 step, bs=16; the first step also runs "
         "MIOpen's solver search, whose kernels are in the totals).  CSV: `profiles/%s_bench_kernel_stats.csv`.\n" % tag)
L.append("Total GPU kernel time %.1f ms = %.2f ms/step (un-profiled wall: %.2f ms/step).\n" % (total_ns / 1e6, total_ns / 1e6 / steps, full["ms_per_step"]))
L.append("## libumr_hip.so kernels (rocprofv3 averages)\n")
L.append("| kernel | calls/step | avg us | ms/step |\n|---|---|---|---|")
osum = 0.0
for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    osum += ms
    L.append("| `%s` | %.1f | %.1f | %.3f |" % (short(r["Name"]), int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, ms))
L.append("\nSum: %.2f ms/step of %.2f (%.0f %%); the rest is the network (MIOpen fp32 convolutions, batch-norm, GEMMs), Adam "
         "and torch elementwise ops.\n" % (osum, total_ns / 1e6 / steps, 100 * osum / (total_ns / 1e6 / steps)))
L.append("## Top 15 kernels overall\n")
L.append("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:15]:
    L.append("| `%s` | %s | %.2f | %.1f | %.1f |" % (short(r["Name"])[:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
rf = full["roofline"]
L.append("\n## bench.py lines of the same build (un-profiled)\n")
L.append("Full step (`profiles/%s_bench_full.json`): **%.0f images/s**, %.2f ms/step; roofline (texture-render raster "
         "backward, HIP events recorded by the library on the launch stream): avg %.1f us/launch, %.0f GB/s algorithmic = "
         "%.1f %% of 8 TB/s, PMC traffic %.1f MB/launch vs %.1f MB algorithmic; cpu_baseline %.2f images/s on %d threads.\n"
         % (tag, full["value"], full["ms_per_step"], rf["avg_us"], rf["achieved"], 100 * rf["frac"],
            (rf["traffic"] or 0) / 1e6, rf["alg_bytes_per_launch"] / 1e6, full["cpu_baseline"]["value"], full["cpu_baseline"]["cores"]))
L.append("Hot path only, `--model 0` (`profiles/%s_bench_hotpath_only.json`): **%.0f images/s**, %.2f ms/step.\n" % (tag, hot["value"], hot["ms_per_step"]))
gpath = os.path.join(SRC, "bench_hotpath_graph.json")
if os.path.exists(gpath):
    hg = json.load(open(gpath))
    L.append("Hot path replayed from one HIP graph, `--model 0 --graph 1` (`profiles/%s_bench_hotpath_graph.json`): **%.0f images/s**, "
             "%.2f ms/step (host enqueue %.3f ms/step).\n" % (tag, hg["value"], hg["ms_per_step"], hg["config"]["host_enqueue_ms_per_step"]))
for name, key in (("textured forward", "forward_kernel"), ("silhouette forward", "silhouette_forward"), ("silhouette backward", "silhouette_backward")):
    k = rf.get(key)
    if k and k.get("avg_us"):
        L.append("%s: avg %.1f us/launch, %.0f GB/s algorithmic = %.1f %% of 8 TB/s (%.1f MB/launch).\n"
                 % (name, k["avg_us"], k["achieved"], 100 * k["frac"], k["alg_bytes_per_launch"] / 1e6))
L.append("train_s2 sequence, `--workload s2` (`profiles/%s_bench_s2.json`, 10 steps): %.0f images/s, %.1f ms/step.\n" % (tag, s2["value"], s2["ms_per_step"]))


def _load(name):
    pth = os.path.join(SRC, name)
    try:
        return json.load(open(pth)) if os.path.exists(pth) and os.path.getsize(pth) > 0 else None
    except ValueError:
        return None


fg, c4, dd, ev = _load("bench_full_graph.json"), _load("bench_s2_cfg4.json"), _load("bench_ddp1.json"), _load("eval_bench.json")
if fg:
    L.append("Whole training step replayed from ONE HIP graph, `--graph 1` (`profiles/%s_bench_full_graph.json`): hip_graph = %s, "
             "**%.0f images/s**, %.2f ms/step, host enqueue %.2f ms/step (eager line above: %.2f ms of host enqueue per %.2f ms step) -- "
             "the step is bound by its GPU work, not by the launches.\n"
             % (tag, fg["config"]["hip_graph"], fg["value"], fg["ms_per_step"], fg["config"]["host_enqueue_ms_per_step"],
                full["config"]["host_enqueue_ms_per_step"], full["ms_per_step"]))
if c4:
    L.append("BASELINE configs[3] shape, `--workload s2 --image-size 512 --subdivide 4` (`profiles/%s_bench_s2_cfg4.json`; bs 16, "
             "IS = 1024, 5120 faces, K = 8): **%.0f images/s**, %.1f ms/step; textured raster backward %.0f us/launch (N = 128 views), "
             "%.1f %% of 8 TB/s algorithmic.\n" % (tag, c4["value"], c4["ms_per_step"], c4["roofline"]["avg_us"] or 0, 100 * (c4["roofline"]["frac"] or 0)))
if dd:
    cfg = dd["config"]
    L.append("1-rank RCCL run, `--force-ddp 1` (`profiles/%s_bench_ddp1.json`): %.0f images/s; the model's %.0f MB of gradients "
             "all-reduced stand-alone in %d buckets of %d MB: %.2f ms (one rank: RCCL's launch + local-copy floor).\n"
             % (tag, dd["value"], cfg.get("allreduce_bytes", 0) / 1e6, cfg.get("ddp_buckets", 0), cfg.get("ddp_bucket_mb", 0),
                cfg.get("allreduce_ms_standalone", float("nan"))))
if ev:
    L.append("Evaluation path (BASELINE configs[4], `profiles/%s_eval_bench.json`): %d synthetic pairs x 2 "
             "directions x %d keypoints -- flow mode %.0f pairs/s (%.1f us/pair), cam mode %.0f pairs/s (%.1f us/pair), PCK counters on "
             "the device.\n" % (tag, ev["pairs"], ev["keypoints"], ev["flow"]["pairs_per_s"], ev["flow"]["us_per_pair"],
                                 ev["cam"]["pairs_per_s"], ev["cam"]["us_per_pair"]))
bk = [r for r in ours if ("k_raster_backward_fm_ag<1" in r["Name"] or "k_raster_backward_fm<1, false, true" in r["Name"])]
tsum = os.path.join(SRC, "raster_trace_summary.json")
if os.path.exists(tsum):
    shutil.copyfile(tsum, os.path.join(P, tag + "_raster_trace_summary.json"))
    tj = json.load(open(tsum))
    kk = [k for k in tj if "k_raster_backward_fm_ag<1" in k] or [k for k in tj if "k_raster_backward_fm<1, false, true" in k]
    if kk:
        L.append("HIP-event average (profile pass = steps 41-45 of the un-profiled bench run) vs rocprofv3 over the same five "
                 "steps of the profiled run of the same command for `%s`: %.1f us vs %.1f us (%.1f us over all 45 steps); "
                 "`profiles/%s_raster_trace_summary.json`.  The two are different processes on different training trajectories "
                 "(float-atomic summation order differs, so the meshes of steps 41-45 are not the same meshes) and at "
                 "different clock states; over this round's refreshes the ratio of the two ranged 0.94 - 1.27.\n"
                 % (kk[0].split("(")[0], rf["avg_us"], tj[kk[0]].get("avg_us_profile_pass", tj[kk[0]].get("avg_us_last10steps", 0.0)),
                    tj[kk[0]]["avg_us_all"], tag))
        hs = os.path.join(SRC, "raster_hot_stats.json")
        if os.path.exists(hs):
            shutil.copyfile(hs, os.path.join(P, tag + "_raster_hot_stats.json"))
            hj = json.load(open(hs))
            hk = [k for k in hj if "k_raster_backward_fm_ag<1" in k] or [k for k in hj if "k_raster_backward_fm<1, false, true" in k]
            if hk:
                L.append("Where both see the same work -- the hot path alone (`--model 0`: the same scene every step) -- they "
                         "agree: HIP events %.1f us (`profiles/%s_bench_hotpath_only.json`) vs rocprofv3 %.1f us over %d "
                         "dispatches (`profiles/%s_raster_hot_stats.json`).\n"
                         % (hot["roofline"]["avg_us"], tag, hj[hk[0]]["avg_us"], hj[hk[0]]["calls"], tag))
elif bk:
    L.append("HIP-event average vs rocprofv3 average for `k_raster_backward_fm<1, false, true, ...>`: %.1f us vs %.1f us.\n"
             % (rf["avg_us"], float(bk[0]["AverageNs"]) / 1e3))
L.append("## HBM traffic (PMC, `tools/collect_traffic.sh`, separate FETCH_SIZE / WRITE_SIZE passes)\n")
c = tr["calibration"]
L.append("Calibration: `%s` reads %d known bytes with dword loads; FETCH_SIZE reported %.0f KB => correction factor %.3f "
         "(the guide's 1/2 under-count on gfx950 confirmed for this access width). WRITE_SIZE taken as is.\n"
         % (c["kernel"], c["known_read_bytes"], c["FETCH_SIZE_KB"], c["fetch_correction_factor"]))
L.append("| kernel (N=16 launches inside bench) | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes/launch (corrected) |\n|---|---|---|---|")
for k, v in tr["kernels"].items():
    L.append("| `%s` | %.0f | %.0f | %.1f MB |" % (short(k), v["FETCH_SIZE_KB"], v["WRITE_SIZE_KB"], v["hbm_bytes_per_launch"] / 1e6))
L.append("\n## VALU roofline of the same launches (SQ PMC passes of the bench command, build %s)\n" % tr.get("build_id"))
L.append("Peak = 157.3 TFLOP/s fp32 vector = 1228.9 G wave64-instructions/s.  `lane use` = SQ_THREAD_CYCLES_VALU / (64 x "
         "SQ_ACTIVE_INST_VALU); wait / issue-stall / active = the three disjoint buckets of a wave's life.\n")
L.append("| kernel | us (PMC pass) | issued VALU wave-instr | SALU wave-instr | lane use | % of VALU peak | wait | issue stall | active |\n|---|---|---|---|---|---|---|---|---|")
for k, v in tr["kernels"].items():
    u = v.get("valu")
    if u:
        L.append("| `%s` | %.1f | %.1f M | %.1f M | %.3f | %.1f | %.2f | %.2f | %.2f |"
                 % (short(k)[:60], u["kernel_us_in_pmc_pass"], u["issued_wave_instr"] / 1e6, (u.get("salu_wave_instr") or 0) / 1e6,
                    u.get("lane_use") or 0, 100 * u["frac_of_peak"], u.get("wave_wait_frac") or 0, u.get("wave_issue_stall_frac") or 0,
                    u.get("wave_active_frac") or 0))
L.append("\n## Kernel timings and SQ counters\n")
L.append("`profiles/%s_microbench.log`: kernel-only HIP-event averages in us per launch [forward, backward] at N=16 / N=128 and at the "
         "configs[3] raster shape (`tools/sweep_fm.py`).\n" % tag)
pj = os.path.join(SRC, "pmc", "pmc_summary.json")
if os.path.exists(pj):
    pm = json.load(open(pj))
    L.append("`profiles/%s_pmc_step_kernels.json` (`tools/r4/pmc_passes.py`: one counter group per run over the four raster launches of a "
             "train_s1 step, N = 16 / 32, kernel-only driver):\n" % tag)
    L.append("| kernel | VALU M | SALU M | VALU busy (ACTIVE_INST_VALU x 4 / SIMD-cycles) | quad-cycles per VALU instr | wait | issue stall | TCC hit | scalar-cache hit | L1->L2 read latency (cycles) |\n|---|---|---|---|---|---|---|---|---|---|")
    for k, c in pm["kernels"].items():
        if "k_raster" not in k:
            continue
        dur = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
        L.append("| `%s` | %.1f | %.1f | %.2f | %.3f | %.2f | %.2f | %s | %s | %s |" % (
            short(k)[:56], c.get("SQ_INSTS_VALU", 0) / 1e6, c.get("SQ_INSTS_SALU", 0) / 1e6,
            c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(dur * 1024, 1), c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_INSTS_VALU", 1), 1),
            c.get("SQ_WAIT_ANY/WAVE_CYCLES", 0), c.get("SQ_WAIT_INST_ANY/WAVE_CYCLES", 0), c.get("tcc_hit_rate"), c.get("scalar_cache_hit_rate"),
            c.get("avg_l1_to_l2_read_latency_cycles")))
sk = os.path.join(SRC, "step_kernels.jsonl")
if os.path.exists(sk):
    L.append("\nKernel-only timings of those launches (`tools/r4/step_kernels.py`, library-owned HIP events, us per launch):\n\n```")
    L.extend(l.strip() for l in open(sk) if l.startswith("{"))
    L.append("```")
for nm, label in (("bench_full_two_renders.json", "train_s1"), ("bench_s2_two_renders.json", "train_s2")):
    t = _load(nm)
    o = full if label == "train_s1" else s2
    if t:
        L.append("\n%s, mask render as the alpha channel of the textured render (default) vs the reference's two renders (`--share-mask-render 0`, "
                 "`profiles/%s_%s`): %.1f vs %.1f images/s.\n" % (label, tag, nm, o["value"], t["value"]))
open(os.path.join(P, tag + "_SUMMARY.md"), "w").write("\n".join(L) + "\n")
print("wrote", os.path.join(P, tag + "_SUMMARY.md"))
