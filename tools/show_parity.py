import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
for k in ("parity","parity_last_rank","shard_window_last_rank"): print(k, d.get(k))
print(d["config"]["per_rank"])
