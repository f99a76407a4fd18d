import json,sys
d=json.loads(sys.stdin.read()); print(round(d["wall_ms_per_iteration_incl_host"],2), round(d["linearise_ms"],2), round(d["backward_ms"],2), round(d["rollouts_ms"],2))
