import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    print(f, round(d["value"]), round(d["ms_per_step"],1), d["all_verified"])
    print("   ", {k:round(v,1) for k,v in d["kernel_ms_per_step"].items() if not k.startswith("host")})
