"""The energy ledger of the shipped inference step: J per image = socket power / images per second of the 3-chain
pipeline (bench.py, configs[1], 8 images x 3 chains), measured again with ONE ingredient taken out at a time --

  * whole launch classes dropped (`bench.py --skip-calls`): the Winograd transforms, crop_and_resize, the proposal / detection stages, the
    splitter, every frcnn_gemm_h2 launch;
  * the dominant kernel's slab loop with one ingredient compiled out (csrc/gemm_h2.hip, -DFRCNN_ABLATION cfgs 50-55 forced onto every
    frcnn_gemm_h2 launch, against cfg 31 forced the same way): the MFMAs, the LDS fragment reads, the L2 -> LDS slab loads;
  * the idle chip (context alive, nothing running).

The socket sits at its power limit in every pipeline run, so time follows joules: E(without X) = P / rate(without X), and
E(X) = E(all) - E(without X).  Every ablated run computes WRONG results by construction (each line says so); operands stay realistic
(one un-ablated pass fills every buffer first; the no-fragment kernel keeps the first 16-k group of every block).

    python scratch/energy_ledger.py [--steps 30] [--out gpurun_out/r06_energy_ledger.txt]
"""
import json, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 30
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "gpurun_out", "r06_energy_ledger.txt")
COMMON = ["--no-cpu-baseline", "--no-other-configs", "--no-f32-variant", "--profile-steps", "0", "--steps", str(steps), "--warmup", "3"]


def run(label, extra, ablation=False):
    cmd = [sys.executable] + ([os.path.join(HERE, "bench_ablation.py"), "--"] if ablation else [os.path.join(ROOT, "bench.py")]) + COMMON + extra
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("%-44s FAILED\n%s" % (label, r.stderr[-800:]), flush=True)
        return None
    j = json.loads(line[-1])
    tel = j.get("telemetry") or {}
    rec = dict(label=label, ips=j["value"], ms=j["ms_per_step"], w=tel.get("socket_w"), mhz=tel.get("sclk_mhz"), wall=time.time() - t0)
    rec["j"] = rec["w"] / rec["ips"] if rec["w"] else None
    print("%-44s %8.1f img/s %8.3f ms/step %7s W %6s MHz %7s J/image   (%.0f s)" % (
        label, rec["ips"], rec["ms"], rec["w"], rec["mhz"], "%.3f" % rec["j"] if rec["j"] else None, rec["wall"]), flush=True)
    return rec


def idle_power(seconds=4.0):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tf-faster-rcnn_amd"), os.path.join(ROOT, "tf-faster-rcnn_amd", "lib")]
    import torch, bench
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda").item()
    tel = bench.Telemetry(bench.Telemetry.pci_of(torch.device("cuda", 0))).run(lambda: time.sleep(seconds))
    return tel


recs = {}
runs = [
    ("shipped", [], False),
    ("shipped (repeat)", [], False),
    ("no Winograd transforms", ["--skip-calls", "frcnn_winograd_input_transform,frcnn_winograd_output_transform,frcnn_winograd7_input_transform,frcnn_winograd7_output_transform"], False),
    ("no crop_and_resize", ["--skip-calls", "frcnn_crop_and_resize"], False),
    ("no proposal / detection stages", ["--skip-calls", "frcnn_proposal_layer,frcnn_detect_post,frcnn_rpn_softmax,frcnn_softmax_rows"], False),
    ("no frcnn_gemm_h2 launches", ["--skip-calls", "frcnn_gemm_h2"], False),
    ("no GEMM / conv launches at all", ["--skip-calls", "frcnn_gemm_h2,frcnn_gemm_x3,frcnn_conv2d,frcnn_gemm_batched"], False),
    ("ONLY the GEMM / conv launches", ["--skip-calls", "frcnn_winograd,frcnn_crop_and_resize,frcnn_proposal_layer,frcnn_detect_post,frcnn_rpn_softmax,frcnn_softmax_rows,frcnn_h2_split,frcnn_maxpool,frcnn_spatial_mean"], False),
    ("h2 launches forced to cfg 31 (ablation build)", ["--hip", "H2_TILE_CFG=31"], True),
    ("cfg 50: no MFMAs", ["--hip", "H2_TILE_CFG=50"], True),
    ("cfg 51: 1/8 of the LDS fragment reads", ["--hip", "H2_TILE_CFG=51"], True),
    ("cfg 52: no slab loads after the prologue", ["--hip", "H2_TILE_CFG=52"], True),
    ("cfg 53: loads + epilogue only", ["--hip", "H2_TILE_CFG=53"], True),
    ("cfg 54: MFMAs + epilogue only", ["--hip", "H2_TILE_CFG=54"], True),
    ("cfg 55: epilogue + loop skeleton only", ["--hip", "H2_TILE_CFG=55"], True),
]
only = [a for a in sys.argv[1:] if a.startswith("only=")]
for label, extra, abl in runs:
    if only and not any(tok in label for tok in only[0][5:].split("|")):
        continue
    recs[label] = run(label, extra, abl)
idle = idle_power()
lines = []
lines.append("# energy ledger of the shipped inference step: configs[1], 8 images x 3 chains, hipGraph replay, %d steps per run (scratch/energy_ledger.py)" % steps)
lines.append("# E = socket W / (images / s).  Ablated runs compute wrong results by construction; operands stay realistic.")
lines.append("%-46s %9s %9s %8s %7s %9s" % ("run", "img/s", "ms/step", "W", "MHz", "J/image"))
for label, r in recs.items():
    if r:
        lines.append("%-46s %9.1f %9.3f %8s %7s %9s" % (label, r["ips"], r["ms"], r["w"], r["mhz"], "%.3f" % r["j"] if r["j"] else "-"))
lines.append("idle (context alive, nothing running): %s W at %s MHz" % (idle and idle.get("socket_w"), idle and idle.get("sclk_mhz")))
base = recs.get("shipped")
if base and base["j"]:
    E = base["j"]
    lines.append("")
    lines.append("## ledger: J per image of the shipped step = %.3f (%.0f W / %.1f images/s)" % (E, base["w"], base["ips"]))

    def term(name, without):
        r = recs.get(without)
        if r and r["j"]:
            lines.append("  %-58s %6.3f J  (%4.1f %%)   [shipped - '%s']" % (name, E - r["j"], 100.0 * (E - r["j"]) / E, without))
    term("Winograd transforms (64 launches per step)", "no Winograd transforms")
    term("crop_and_resize", "no crop_and_resize")
    term("proposal layer + per-class NMS + softmaxes", "no proposal / detection stages")
    term("every frcnn_gemm_h2 launch", "no frcnn_gemm_h2 launches")
    term("every GEMM / conv launch", "no GEMM / conv launches at all")
    g = recs.get("ONLY the GEMM / conv launches")
    if g and g["j"]:
        lines.append("  %-58s %6.3f J  (%4.1f %%)" % ("the GEMM / conv launches alone (everything else dropped)", g["j"], 100.0 * g["j"] / E))
    if idle and idle.get("socket_w"):
        e_idle = idle["socket_w"] / base["ips"]
        lines.append("  %-58s %6.3f J  (%4.1f %%)   [idle W x seconds per image]" % ("static share at the idle clock", e_idle, 100.0 * e_idle / E))
    b31 = recs.get("h2 launches forced to cfg 31 (ablation build)")
    if b31 and b31["j"]:
        lines.append("")
        lines.append("## inside frcnn_gemm_h2's slab loop (every h2 launch forced to the 128 x 128 light-boundary kernel: %.3f J per image)" % b31["j"])

        def kterm(name, without):
            r = recs.get(without)
            if r and r["j"]:
                lines.append("  %-58s %6.3f J  (%4.1f %% of the shipped step)   [cfg 31 - '%s']" % (name, b31["j"] - r["j"], 100.0 * (b31["j"] - r["j"]) / E, without))
        kterm("the MFMAs (3 x v_mfma_f32_32x32x16_f16 per product)", "cfg 50: no MFMAs")
        kterm("7/8 of the LDS fragment reads (ds_read_b128)", "cfg 51: 1/8 of the LDS fragment reads")
        kterm("the slab loads L2 -> LDS (+ L2 / fabric / HBM behind them)", "cfg 52: no slab loads after the prologue")
        kterm("MFMAs + fragment reads together", "cfg 53: loads + epilogue only")
        kterm("slab loads + fragment reads together", "cfg 54: MFMAs + epilogue only")
        r55 = recs.get("cfg 55: epilogue + loop skeleton only")
        if r55 and r55["j"]:
            lines.append("  %-58s %6.3f J  (%4.1f %% of the shipped step)" % ("what remains: epilogues, barriers, loop skeleton + all other kernels", r55["j"], 100.0 * r55["j"] / E))
os.makedirs(os.path.dirname(out_path), exist_ok=True)
open(out_path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
