"""Print the figures of one bench_detail.json (the full record `bench.py` writes next to its < 4 KB stdout line) that DESIGN.md / README.md / profiles/README.md quote.
usage: summarize_bench.py [bench_detail.json]"""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "bench_detail.json"))
c = d["config"]
print("value %.2f M rays/s  %.2f ms/step  spr %.2f  %.3f G samples/s" % (d["value"] / 1e6, d["ms_per_step"], c.get("samples_per_ray", 0), c.get("samples_per_s", 0) / 1e9))
r = d.get("roofline")
if r: print("roofline frac %.4f  avg launch %.4f ms  samples/launch %.3f M  launches %d  share %.3f  traffic/launch %.3f GB  achieved %.0f GB/s" % (r["frac"], r["avg_launch_ms"], r["samples_per_launch"] / 1e6, r["launches"], r["field_kernel_share_of_serial_step"], (r["traffic"] or 0) / 1e9, r["achieved"]))
if "render_views1" in d: v = d["render_views1"]; print("views1 %.2f M rays/s  %.2f ms/view  spr %.1f" % (v["value"] / 1e6, v["ms_per_view"], v["samples_per_ray"]))
for key in ("train", "train_refyaml"):
    if key in d:
        t = d[key] if key != "train" else d[key][d[key]["dtype"]]
        ro = t["roofline"]
        print("%s %.3f ms/step  kept %.3f M  marched %.3f M  frac %.3f  traffic %s GB (%s x algorithmic)" % (key, t["ms_per_step"], t["rendering_samples_per_step"] / 1e6, t["marched_samples_per_step"] / 1e6, ro["frac"], "%.2f" % (ro["traffic"] / 1e9) if ro["traffic"] else "-", "%.2f" % ro["traffic_over_algorithmic"] if ro["traffic_over_algorithmic"] else "-"))
        if "kernels" in t: print("   " + "  ".join("%s %.3f" % (k, v["ms_per_step"]) for k, v in t["kernels"].items()))
if "config2" in d:
    c2 = d["config2"]; rr = c2["render"]
    print("config2 render %.2f ms/view  %.1f M rays/s  field frac %.3f | train %.3f ms (2000 rays)  %.3f ms (8192)" % (rr["ms_per_view"], rr["rays_per_s"] / 1e6, rr.get("roofline", {}).get("frac", 0), c2["train"]["ms_per_step"], c2["train_8192"]["ms_per_step"]))
if "score256" in d: s = d["score256"]; print("score256 %.1f ms/pass  %.2f M rays/s  spr %.1f  %.2f G samples/s" % (s["ms_per_pass"], s["rays_per_s"] / 1e6, s.get("samples_per_ray_rank0", 0), s.get("samples_per_s_rank0", 0) / 1e9))
if "score256_shard8" in d: print("shard8 %.2f ms/pass  ratio to full/8 %.2f" % (d["score256_shard8"]["ms_per_pass"], d["score256_shard8"]["ratio_to_full_over_8"]))
if "cpu_baseline" in d: b = d["cpu_baseline"]; print("cpu %.0f rays/s (%d threads)  %s" % (b["value"], b["cores"], b["cpu_model"]))
if "bench_parity" in d: print("parity ok %s  max abs %s" % (d["bench_parity"]["ok"], d["bench_parity"]["max_abs"]))
for k, v in c.get("standin_training", {}).items(): print("   standin", k, "loss %.3f occupied %d/%d" % (v["loss_last"], v["occupied_cells"], v["cells"]))
