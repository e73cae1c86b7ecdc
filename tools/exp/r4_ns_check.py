import json,sys
for f in sys.argv[1:]:
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "no json", e); continue
    a=d["also"]["c4_c5_shard"]
    print(f, {k: round(v.get("kernel_ms", v.get("call_ms", 0)),1) for k,v in a.items() if isinstance(v, dict)})
