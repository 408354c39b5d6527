"""prints the figures of a bench.py JSON line that the docs quote.   python tools/show_bench.py bench.json"""
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "plan", d["config"]["rows_per_pass"], "x", d["config"]["passes_in_flight_per_gpu"])
print("roofline frac", r["frac"], "live us", r["avg_launch_us"], "in situ us", r["in_situ_us"], "alone", r["alone"]["us"], r["alone"]["frac"],
      "window", (r.get("in_flight") or {}).get("decode_window_hbm"))
if "value_batch16" in d:
    b = d["value_batch16"]
    print("value_batch16", b["value"], "frac", b["roofline"]["frac"], "live us", b["roofline"]["avg_launch_us"], "in situ", b["roofline"]["in_situ_us"], "identical", b["tokens_identical_to_value_run"])
for k in ("job_30min", "vad_mix", "config4", "cpu_baseline", "align_stage"):
    if k in d:
        print(k, json.dumps(d[k])[:400])
print("more", json.dumps(d.get("roofline_more"))[:700])
