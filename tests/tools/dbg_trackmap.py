import numpy as np, sys
sys.path.insert(0,'/root/repo')
from ptam_cg_amd._lib import load
from tests.oracle_lib import load_oracle
import tests.test_gpu_trackmap as T
hip, o = load(), load_oracle()
for name,(ck,ok) in T.CASES.items():
    ck=dict(ck); counts=ck.pop("counts")
    res,it,case=T._run_hip(hip,counts,ok,**ck); ref=T._run_ref(o,counts,ok,**ck)
    rit=ref["iteration_set"]
    same = len(it)==len(rit) and all(np.array_equal(it[k],rit[k]) for k in ("point","level","found","did_subpix","outlier"))
    f=(it["found"]==1) if same else None
    dv=np.abs(it["v2_found"][f]-rit["v2_found"][f]).max() if same and f.any() else -1
    wi=np.argmax(np.abs(it["v2_found"][f]-rit["v2_found"][f]).max(1)) if same and f.any() else -1
    print(name, "discrete_same",same,"v2maxdiff",dv, "sub" , it["did_subpix"][f][wi] if same and f.any() else None, "pose diff",np.abs(res["pose"]-ref["pose"]).max(), "depth", res["depth_n"], ref["depth"][2], res["depth_sum"]-ref["depth"][0],
          "counts", list(res["attempted"])==ref["attempted"], list(res["found"])==ref["found"], res["n_meas"], ref["n_meas"], bool(res["did_coarse"]), ref["did_coarse"])
print("---- composed through the HIP library's batch calls vs the chain / the oracle ----")
for name in ("no_coarse", "coarse_and_chop"):
    ck,ok = T.CASES[name]; ck=dict(ck); counts=ck.pop("counts")
    res,it,case=T._run_hip(hip,counts,ok,**ck); ref=T._run_ref(o,counts,ok,**ck); refh=T._run_ref(hip,counts,ok,**ck)
    f=it["found"]==1
    a=np.abs(it["v2_found"][f]-ref["iteration_set"]["v2_found"][f]).max(1)
    b=np.abs(it["v2_found"][f]-refh["iteration_set"]["v2_found"][f]).max(1)
    c=np.abs(refh["iteration_set"]["v2_found"][f]-ref["iteration_set"]["v2_found"][f]).max(1)
    print(name,"chain-oracle",a.max(),"chain-composedHIP",b.max(),"composedHIP-oracle",c.max(), "n bad", (a>1e-9).sum(), "levels of bad", it["level"][f][a>1e-9][:10], "slots", np.flatnonzero(f)[a>1e-9][:10])
