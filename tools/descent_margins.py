import sys, os, math, json
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0]=[ROOT, os.path.join(ROOT,"nextgen-uia_amd"), os.path.join(ROOT,"tools")]
import descent_check as D
for rep in range(3):
    big = D.run_hip(64, 60, 1e-3)
    small = D.run_hip(8, 14, 1e-3)
    ref = D.run_oracle(8, 14, 1e-3, threads=16)
    cos, ratio = D.alignment(small, ref)
    l0, l1 = D.oracle_loss(None, 8, 14, threads=16), D.oracle_loss(small["state"], 8, 14, threads=16)
    print(json.dumps({"rep": rep, "B64_first": round(big["losses"][0],3), "B64_min_last5_over_first": round(min(big["losses"][-5:])/big["losses"][0],3),
                      "B8_hip_last_over_first": round(small["losses"][-1]/small["losses"][0],3), "B8_ref_last_over_first": round(ref["losses"][-1]/ref["losses"][0],3),
                      "first6_max_rel": round(max(abs(a-b)/b for a,b in zip(small["losses"][:6], ref["losses"][:6])),4), "cos": round(cos,3), "ratio": round(ratio,3),
                      "oracle_loss_untrained": round(l0,3), "oracle_loss_hip_trained": round(l1,3), "hip_last": round(small["losses"][-1],3)}), flush=True)
