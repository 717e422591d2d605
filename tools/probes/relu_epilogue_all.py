import subprocess, sys
from cnn_amd.stacks import conv_geometries
seen=set()
for name,B in (("vgg11",128),("resnet18",64)):
    for (Ci,H,W,Co,k,s,pad) in conv_geometries(name):
        case=(B,Ci,H,W,Co,k,s,pad)
        if case in seen or Ci<16: continue
        seen.add(case)
        out=subprocess.run([sys.executable,"tools/probes/relu_epilogue.py"]+[str(v) for v in case],capture_output=True,text=True).stdout
        rows={}
        for l in out.splitlines():
            if "prep" in l or "us" not in l: continue
            f=l.split()
            rows[f[0]]=(float(f[1]), f[-1].split("|")[0])
        def r(a,b):
            return f"{rows[a][0]:8.1f} -> {rows[b][0]:8.1f} ({rows[b][0]/rows[a][0]:.2f}x)" if a in rows and b in rows else "n/a"
        print(case, "fwd:", r("fwd","fwd+relu"), "| dgrad:", r("dgrad","dgrad+relu"), "|", rows.get("fwd+relu",("",""))[1], rows.get("dgrad+relu",("",""))[1], flush=True)
