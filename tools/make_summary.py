"""Copy the outputs of tools/refresh_fast.sh (+ tools/refresh_slow.sh when present) from gpurun_out/ into profiles/ and write
profiles/<tag>_SUMMARY.md.  Every figure is stamped with the build id of the libumr_hip.so that produced it (bench lines:
config.lib_build_id; traffic.json / refresh_fast: build_id); a file measured on ANOTHER build than the tree's sources
(umr_amd.build.source_hash()) is REFUSED -- the summary never mixes builds.
Usage: python tools/make_summary.py <tag>        (e.g. r05)"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from umr_amd import build as B  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
FAST, SLOW, P = os.path.join(ROOT, "gpurun_out", "refresh_fast"), os.path.join(ROOT, "gpurun_out", "refresh_slow"), os.path.join(ROOT, "profiles")
HASH = B.source_hash()
refused = []


def line(path):
    """Last JSON line of a bench output, or None (missing / empty / another build)."""
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        return None
    bid = d.get("config", {}).get("lib_build_id")
    if bid != HASH:
        refused.append("%s (build %s)" % (os.path.relpath(path, ROOT), bid))
        return None
    return d


def keep(src, name):
    shutil.copyfile(src, os.path.join(P, tag + "_" + name))


if open(os.path.join(FAST, "build_id.txt")).read().strip() != HASH:
    raise SystemExit("make_summary: gpurun_out/refresh_fast was measured on build %s, the tree is %s -- run tools/refresh_fast.sh again"
                     % (open(os.path.join(FAST, "build_id.txt")).read().strip(), HASH))
tr = json.load(open(os.path.join(FAST, "traffic", "traffic.json")))
if tr.get("build_id") != HASH:
    raise SystemExit("make_summary: traffic.json is of build %s, the tree is %s" % (tr.get("build_id"), HASH))
full, hot = line(os.path.join(FAST, "bench_full.json")), line(os.path.join(FAST, "bench_hotpath_only.json"))
if not full:
    raise SystemExit("make_summary: no usable bench_full.json: %s" % refused)
keep(os.path.join(FAST, "bench_full.json"), "bench_full.json")
if hot:
    keep(os.path.join(FAST, "bench_hotpath_only.json"), "bench_hotpath_only.json")
keep(os.path.join(FAST, "stats", "t_kernel_stats.csv"), "bench_kernel_stats.csv")
keep(os.path.join(FAST, "traffic", "traffic.json"), "traffic.json")
shutil.copyfile(os.path.join(FAST, "traffic", "traffic.json"), os.path.join(P, "traffic.json"))
slow = {}
for name in ("bench_full_eager", "bench_s2", "bench_s2_cfg4", "bench_ddp1", "bench_full_two_renders", "bench_s2_two_renders"):
    d = line(os.path.join(SLOW, name + ".json"))
    if d:
        slow[name] = d
        keep(os.path.join(SLOW, name + ".json"), name + ".json")
kern = []
if os.path.exists(os.path.join(SLOW, "kernels.jsonl")):
    kern = [json.loads(l) for l in open(os.path.join(SLOW, "kernels.jsonl")) if l.startswith("{")]
    if all(k.get("build") == HASH[:12] for k in kern) and kern:
        keep(os.path.join(SLOW, "kernels.jsonl"), "fixed_scene_kernels.jsonl")
    else:
        refused.append("gpurun_out/refresh_slow/kernels.jsonl"); kern = []
for extra in ("cold_cache.json", "concurrency.json", "kernel_only.log"):
    if os.path.exists(os.path.join(SLOW, extra)) and os.path.getsize(os.path.join(SLOW, extra)) > 0 and kern:
        keep(os.path.join(SLOW, extra), extra)

keep(os.path.join(FAST, "steady_kernel_stats.csv"), "bench_kernel_stats_steady.csv")
rows = list(csv.DictReader(open(os.path.join(FAST, "steady_kernel_stats.csv"))))
tot = json.load(open(os.path.join(FAST, "steady_totals.json")))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("(RasterArgs)", "")[:96]
ours = [r for r in rows if short(r["Name"]).startswith(("k_", "umr_k"))]
L = ["# Round %s profile summary (1x MI355X)\n" % tag.lstrip("r0"),
     "Everything here was measured on ONE build of `libumr_hip.so`: **%s** (= `umr_amd.build.source_hash()` of the tree). Produced on the "
     "GPU box by `tools/refresh_fast.sh` (kernel trace, PMC passes, default line; re-run after the last kernel change) and "
     "`tools/refresh_slow.sh` (other workloads, fixed-scene kernel timings), copied here by `tools/make_summary.py`, which refuses "
     "files of another build.\n" % HASH]
if refused:
    L.append("Refused (another build): %s.\n" % "; ".join(refused))
c, rf = full["config"], full["roofline"]
L.append("## bench.py, default command (`profiles/%s_bench_full.json`)\n" % tag)
L.append("**%.0f images/s**, %.2f ms/step -- train_s1, bs 16, whole step (distance transform, MeshNet, raster + losses, backward, Adam) replayed "
         "from ONE HIP graph (`hip_graph: %s`); enqueueing one eager step costs the host %.1f ms (`eager_host_enqueue_ms_per_step`), a graph "
         "replay %.2f. Render-and-compare step alone, same process (`hot_path_*`): **%.0f images/s**, %.2f ms/step. CPU path "
         "(`cpu_baseline`, oracle on %d threads): %.2f images/s.\n"
         % (full["value"], full["ms_per_step"], c["hip_graph"], c.get("eager_host_enqueue_ms_per_step") or float("nan"),
            c["host_enqueue_ms_per_step"], c.get("hot_path_images_per_s") or float("nan"), c.get("hot_path_ms_per_step") or float("nan"),
            full["cpu_baseline"]["cores"], full["cpu_baseline"]["value"]))
L.append("`roofline` (library-owned HIP events around each raster main kernel).  `avg us` / `%% of 8 TB/s` are the line's DETERMINISTIC figures: "
         "the kernel on the frozen captures of a training step's own geometry (%s; 20 launches each, mean over the scenes) -- they repeat "
         "from run to run; `live us` is what THIS run's trajectory rendered at its profile pass (+-40 %% between runs of one build).  PMC "
         "columns: the step's own launches (`profiles/traffic.json`).\n" % rf.get("source", "-"))
L.append("| launch | avg us | live us | algorithmic MB | % of 8 TB/s | PMC HBM MB | VALU wave-instr | lane use | VALU issue / peak | VALU busy | wave wait |\n|---|---|---|---|---|---|---|---|---|---|---|")
tk = {short(k): v for k, v in tr["kernels"].items()}


def pmc(prefix):
    for k, v in tk.items():
        if k.startswith(prefix):
            return v
    return {}


for name, key, kp in (("shared render's one-pass backward (packed state), N = 16", None, "k_raster_backward_fm_agp<1"),
                      ("textured forward + p2f + visible ids + pool (packed state), N = 16", "forward_kernel", "k_raster_forward<1"),
                      ("silhouette forward, N = 16", "silhouette_forward", "k_raster_forward<2"),
                      ("silhouette backward, N = 16", "silhouette_backward", "k_raster_backward_fm_quads<2")):
    k = rf if key is None else rf.get(key, {})
    t = pmc(kp); v = t.get("valu", {})
    if k.get("avg_us"):
        L.append("| %s | %.1f | %s | %.1f | %.1f | %s | %s | %s | %s | %s | %s |" % (
            name, k["avg_us"], "%.1f" % k["live"]["avg_us"] if k.get("live", {}).get("avg_us") else "-", k["alg_bytes_per_launch"] / 1e6, 100 * k["frac"],
            "%.0f" % (t["hbm_bytes_per_launch"] / 1e6) if t else "-", "%.1f M" % (v["issued_wave_instr"] / 1e6) if v else "-",
            "%.2f" % v["lane_use"] if v.get("lane_use") else "-", "%.3f" % v["frac_of_peak"] if v else "-",
            "%.2f" % v["busy_frac"] if v.get("busy_frac") is not None else "-",
            "%.2f" % v["wave_wait_frac"] if v.get("wave_wait_frac") is not None else "-"))
L.append("\nSummed raster main kernels: %.0f us per step (%s launches).\n" % (rf.get("raster_kernels_us_per_step", float("nan")), rf.get("raster_launches_per_step")))
if rf.get("scenes"):
    L.append("## Frozen scenes, kernel only (`roofline.scenes` of the line above; `tools/scene_times.py`)\n")
    L.append(rf.get("scenes_note", "") + "\n")
    for sn, d_ in rf["scenes"].items():
        L.append("* %s: %s" % (sn, ", ".join("%s %.1f" % kv for kv in d_.items())))
    L.append("")
ss = os.path.join(FAST, "scene_stats", "t_kernel_stats.csv")
if os.path.exists(ss):
    keep(ss, "frozen_scenes_kernel_stats.csv")
    L.append("rocprofv3 `--kernel-trace --stats` of `tools/scene_times.py live_s1_a.npz live_s1_b.npz` (`profiles/%s_frozen_scenes_kernel_stats.csv`; "
             "2 warm-up + 20 timed launches per scene) -- the averages `roofline.avg_us` has to agree with:\n" % tag)
    L.append("| kernel | calls | average us |\n|---|---|---|")
    for r_ in csv.DictReader(open(ss)):
        if "k_raster" in r_["Name"]:
            L.append("| `%s` | %s | %.1f |" % (r_["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70], r_["Calls"], float(r_["AverageNs"]) / 1e3))
    L.append("")
for sc_name in ("live_s1_a", "live_s1_b", "survey_8d"):
    pj = os.path.join(FAST, "pmc_" + sc_name, "pmc_summary.json")
    if os.path.exists(pj):
        keep(pj, "pmc_%s.json" % sc_name)
        pm = json.load(open(pj))
        L.append("Counters of the raster kernels on the frozen scene %s (`profiles/%s_pmc_%s.json`, `tools/pmc_passes.py`, one group per run):\n" % (sc_name, tag, sc_name))
        L.append("| kernel | us in the SQ pass | VALU wave-instr | VALU busy | wave wait | issue stall | L2 hit rate |\n|---|---|---|---|---|---|---|")
        for kn, c_ in pm["kernels"].items():
            if "k_raster" not in kn or "SQ_INSTS_VALU" not in c_:
                continue
            g_ = lambda n, f="%.2f": (f % c_[n]) if c_.get(n) is not None else "-"
            L.append("| `%s` | %s | %.1f M | %s | %s | %s | %s |" % (kn.replace("void ", "")[:60], g_("kernel_us_in_sq2_pass", "%.1f"), c_["SQ_INSTS_VALU"] / 1e6, g_("valu_busy_frac"),
                                                                g_("SQ_WAIT_ANY/WAVE_CYCLES"), g_("SQ_WAIT_INST_ANY/WAVE_CYCLES"), g_("tcc_hit_rate")))
        L.append("")
L.append("## Kernel trace (`rocprofv3 --kernel-trace --stats`)\n")
L.append("Command: `bench.py --steps 10 --warmup 3 --profile-steps 0 --graph 0` (eager: every kernel a dispatch of its own; 13 training steps "
         "from initialisation). `profiles/%s_bench_kernel_stats.csv` is rocprofv3's own table over the whole process -- its first steps hold "
         "MIOpen's solver search (seconds of naive reference convolutions); `profiles/%s_bench_kernel_stats_steady.csv` and the tables below "
         "are the LAST %d steps from the per-dispatch trace: %.2f ms of kernel time per step in %.0f launches (%.2f ms wall per eager step).\n"
         % (tag, tag, tot["steps"], tot["kernel_us_per_step"] / 1e3, tot["launches_per_step"], tot["wall_us_per_step"] / 1e3))
L.append("| libumr_hip.so kernel | calls/step | avg us | us/step |\n|---|---|---|---|")
osum = sum(float(r["UsPerStep"]) for r in ours)
for r in ours:
    if float(r["UsPerStep"]) >= 4.0:
        L.append("| `%s` | %s | %s | %s |" % (short(r["Name"]), r["CallsPerStep"], r["AverageUs"], r["UsPerStep"]))
L.append("\nlibumr_hip.so kernels: %.2f ms/step of %.2f (%.0f %%); the rest is the networks (MIOpen fp32 convolutions, batch-norm, GEMMs), "
         "Adam and torch elementwise kernels (SURVEY 2 row 13: out of scope).\n" % (osum / 1e3, tot["kernel_us_per_step"] / 1e3, 100 * osum / tot["kernel_us_per_step"]))
L.append("Top 10 kernels overall (steady state):\n\n| kernel | calls/step | avg us | us/step | % |\n|---|---|---|---|---|")
for r in rows[:10]:
    L.append("| `%s` | %s | %s | %s | %.1f |" % (short(r["Name"])[:70], r["CallsPerStep"], r["AverageUs"], r["UsPerStep"], 100 * float(r["UsPerStep"]) / tot["kernel_us_per_step"]))
if kern:
    L.append("\n## Fixed SURVEY 8d scene, kernel only (`tools/kernels.py`, library-owned HIP events; `profiles/%s_fixed_scene_kernels.jsonl`)\n" % tag)
    L.append("Identical data every run and every round -- the comparable figures. us per launch.\n")
    for k in kern:
        L.append("* N = %s, camera scale %s: %s" % (k.get("N", 16), k["scale"], ", ".join("%s %.1f" % (n, v) for n, v in k["us_per_launch"].items())))
    L.append("")
L.append("## Other workloads (`tools/refresh_slow.sh`)\n")
for name, what in (("bench_full_eager", "default workload, eager launches (`--graph 0`)"), ("bench_s2", "train_s2, bs 16, K = 8 (`--workload s2`; BASELINE configs[2] per-GPU shape)"),
                   ("bench_s2_cfg4", "configs[3] shape (`--workload s2 --image-size 512 --subdivide 4`: IS 1024, 5120 faces)"),
                   ("bench_ddp1", "1-rank RCCL (`--force-ddp 1`): the whole step incl. the bucket all-reduces from one HIP graph"), ("bench_full_two_renders", "train_s1 with the reference's two renders (`--share-mask-render 0`)"),
                   ("bench_s2_two_renders", "train_s2 with the reference's two renders")):
    d = slow.get(name)
    if not d:
        continue
    r2, c2 = d["roofline"], d["config"]
    extra = ""
    if name == "bench_ddp1":
        extra = "; %.0f MB of gradients all-reduced stand-alone in %d buckets: %.2f ms" % (c2.get("allreduce_bytes", 0) / 1e6, c2.get("ddp_buckets", 0), c2.get("allreduce_ms_standalone", float("nan")))
    L.append("* %s (`profiles/%s_%s.json`): **%.0f images/s**, %.2f ms/step, hip_graph %s, eager host enqueue %s ms/step; raster main kernels %.0f us/step "
             "(%.1f %% of the step), dominant backward %.0f us/launch = %.1f %% of 8 TB/s algorithmic%s."
             % (what, tag, name, d["value"], d["ms_per_step"], c2["hip_graph"], "%.1f" % c2["eager_host_enqueue_ms_per_step"] if c2.get("eager_host_enqueue_ms_per_step") else "-",
                r2.get("raster_kernels_us_per_step") or 0, 100 * (r2.get("raster_kernels_us_per_step") or 0) / 1e3 / d["ms_per_step"],
                r2.get("avg_us") or 0, 100 * (r2.get("frac") or 0), extra))
open(os.path.join(P, tag + "_SUMMARY.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
