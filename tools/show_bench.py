import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/round", d["ms_per_step"], "ppo us/step", d["roofline"]["us_per_step"], "disc us", d["roofline"]["disc_update_us"], "in rounds", d["roofline"]["disc_update_us_in_rounds"])
for k,v in (d.get("variants") or {}).items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a!="config"})
print("cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"])
