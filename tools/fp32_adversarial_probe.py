import sys, os, numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import aerobulk_amd as ab
from oracle import pyoracle as po
from test_gpu_adversarial import adversarial_fields, IN8
for algo,skin,zt,zu,niter in (("coare3p6",True,2.,10.,5),("coare3p6",True,18.,25.,5),("ecmwf",True,2.,10.,6),("coare3p0",False,10.,10.,5),("andreas",False,8.,12.,7)):
    f,which=adversarial_fields(po,algo,skin,zt,zu,niter,20000,900)
    n=f['sst'].size
    f32={k:v.astype(np.float32) for k,v in f.items()}
    f64r={k:v.astype(np.float64) for k,v in f32.items()}
    nt=2 if skin else 1
    osess=po.OracleSession(algo,n,nt,skin)
    for prec in ("f32","f32_storage"):
        with ab.Session(algo,n,1,nt,skin,precision=prec) as s:
            s.set_humidity("sh")
            osess=po.OracleSession(algo,n,nt,skin)
            for jt in range(1,nt+1):
                ref=osess.compute(jt,zt,zu,niter,*[f64r[k] for k in IN8[:6]],rad_sw=f64r['rad_sw'] if skin else None,rad_lw=f64r['rad_lw'] if skin else None)
                got=s.compute(jt,zt,zu,*[f32[k] for k in IN8[:6]],Niter=niter,rad_sw=f32['rad_sw'] if skin else None,rad_lw=f32['rad_lw'] if skin else None)
                line=[]
                for kg,kr in (("QL","ql"),("QH","qh"),("Tau_x","tau_x"),("T_s","t_s")):
                    if kg not in got: continue
                    g=np.asarray(got[kg],dtype=np.float64); r=ref[kr]
                    assert np.all(np.isfinite(g)),(algo,prec,kg)
                    e=np.abs(g-r)
                    line.append(f"{kg} max|err| {e.max():.2e} p99.9 {np.quantile(e,0.999):.2e} (max|ref| {np.abs(r).max():.1f})")
                print(algo,'skin' if skin else 'noskin',zt,prec,'jt',jt,'; '.join(line),flush=True)
