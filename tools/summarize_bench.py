"""Print the figures of one bench.py JSON line that DESIGN.md / README.md / profiles/README.md quote.  usage: summarize_bench.py <file>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = d["config"]
print("value %.2f M rays/s  %.2f ms/step  spr %.1f  %.3f G samples/s" % (d["value"] / 1e6, d["ms_per_step"], c.get("samples_per_ray", 0), c.get("samples_per_s", 0) / 1e9))
r = d.get("roofline")
if r: print("roofline frac %.4f  avg launch %.4f ms  samples/launch %.3f M  launches %d  share %.3f  traffic/launch %.3f GB  achieved %.0f GB/s" % (r["frac"], r["avg_launch_ms"], r["samples_per_launch"] / 1e6, r["launches"], r["field_kernel_share_of_step"], (r["traffic"] or 0) / 1e9, r["achieved"]))
if "render_views1" in d: v = d["render_views1"]; print("views1 %.2f M rays/s  %.2f ms/view  spr %.1f" % (v["value"] / 1e6, v["ms_per_view"], v["samples_per_ray"]))
if "render_random_weights" in d: v = d["render_random_weights"]; print("random weights %.2f M rays/s  %.2f ms/step  spr %.2f  %.3f G samples/s" % (v["value"] / 1e6, v["ms_per_step"], v["samples_per_ray"], v["samples_per_s"] / 1e9))
if "train" in d:
    t = d["train"]; print("train %.3f ms/step  kept %.3f M  marched %.3f M  roofline %.3f" % (t["ms_per_step"], t["rendering_samples_per_step"] / 1e6, t["marched_samples_per_step"] / 1e6, t["roofline"]["frac"]))
    print("   " + "  ".join("%s %.3f" % (k, v["ms_per_step"]) + (" (%.2f)" % v["frac_of_hbm_peak"] if v.get("frac_of_hbm_peak") else "") for k, v in t["kernels"].items()))
if "score256" in d: s = d["score256"]; print("score256 %.1f ms/pass  %.2f M rays/s  spr %.1f  %.2f G samples/s" % (s["ms_per_pass"], s["rays_per_s"] / 1e6, s.get("samples_per_ray_rank0", 0), s.get("samples_per_s_rank0", 0) / 1e9))
if "cpu_baseline" in d:
    b = d["cpu_baseline"]; print("cpu %.0f rays/s (%d thr)  %.0f (1 thr)  BL-1 %.0f / %.0f  %s" % (b["value"], b["threads"], b["value_1thread"] or 0, b["bl1_vanilla_64x64x32"]["rays_per_s"], b["bl1_vanilla_64x64x32"]["rays_per_s_1thread"] or 0, b["cpu_model"]))
for k, v in c.get("standin_training", {}).items(): print("   standin", k, "loss %.3f occupied %d/%d" % (v["loss_last"], v["occupied_cells"], v["cells"]))
