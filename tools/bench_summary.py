"""Short human summary of a bench.py JSON line: python3 tools/bench_summary.py gpurun_out/x.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print("%s | %.2f ms/step | %.1f %s | loss %.4f" % (d["dtype"], d["ms_per_step"], d["value"], d["unit"], d["loss"]))
for k in ("also_split_bf16_x3", "also_f32_mfma"):
    if k in d:
        print("  %-20s %.2f ms/step" % (k, d[k]["ms_per_step"]))
def show(r, ind="  "):
    print(ind + "%s: %.1f TF = %.3f of %.0f (busy %s) | %.1f us/launch x %d = %.2f ms/step%s" % (
        r["kernel"][:60], r["achieved"], r["frac"], r["peak"], r.get("mfma_busy_frac"), r["us_per_launch"], r["launches_per_step"],
        r["ms_per_step"], (" | %.2f us/time step" % r["us_per_time_step"]) if "us_per_time_step" in r else ""))
if "roofline" in d:
    show(d["roofline"])
    if "also" in d["roofline"]:
        show(d["roofline"]["also"])
for e in d.get("encoder_gate_gemm", []):
    print("  L%d %-52s %7.1f us %6.1f TF frac %.3f busy %s %s" % (e["layer"], e["op"][:52], e["us"], e["tflops"], e["frac"], e.get("mfma_busy_frac"),
          ("%.2f us/step" % e["us_per_time_step"]) if e.get("us_per_time_step") else ""))
if "cpu_baseline" in d:
    print("  cpu: %.2f utt/s on %d cores" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
print("  paths:", d["config"]["sequence_op_paths_per_step"])
