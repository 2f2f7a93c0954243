import sys, json, numpy as np, importlib.util, os, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'automatic-speech-recognition_amd'))
from las import _hip
g=json.load(open(os.path.join(ROOT,'tests/golden/reference_host_golden.json')))
spec = importlib.util.spec_from_file_location("mg", os.path.join(ROOT,"tests/golden/make_golden.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
dev='cuda'
for c in g["G5"]:
    V,Tp,D,beam=c["V"],c["Tp"],c["D"],c["beam"]
    toy=m.toy_speller(c["seed"],V,Tp,D); dec_step=int(c["audiolen"]*c["convert_rate"])
    hyps=[dict(ids=[1],lp=np.float32(0),att=np.zeros(Tp,np.float32),st=tuple(np.zeros((1,D),np.float32) for _ in range(2)))]*beam
    sel=[];t=0
    B=dict(score=torch.zeros(1,beam,device=dev),length=torch.zeros(1,beam,dtype=torch.int32,device=dev),nlive=torch.zeros(1,dtype=torch.int32,device=dev),par=torch.zeros(1,beam,dtype=torch.int32,device=dev),tok=torch.zeros(1,beam,dtype=torch.int32,device=dev),osc=torch.zeros(1,beam,device=dev),on=torch.zeros(1,dtype=torch.int32,device=dev),lg=torch.zeros(1,beam,V,device=dev))
    bad=False
    while t<dec_step and len(sel)<beam and not bad:
        N=len(hyps)
        st=np.stack([np.concatenate([h["st"][l] for h in hyps],0) for l in range(2)])
        logits,new,al=m.toy_step(toy,[h["ids"][-1] for h in hyps],np.stack([h["att"] for h in hyps]),st)
        nb = N if t>0 else 1
        cands=[]
        for i in range(nb):
            for v in range(V):
                if t>0 and v==1: continue
                l=logits[i,v]; s=np.float32(hyps[i]["lp"]+l); nrm=np.float32(s/np.float32(len(hyps[i]["ids"])))
                cands.append((nrm,i,l,v,s))
        cands.sort(key=lambda k:(k[0],k[1],k[2],k[3]))
        top=cands[-beam:]
        B["lg"][0,:N]=torch.tensor(logits); B["score"][0,:N]=torch.tensor([float(h["lp"]) for h in hyps]); B["length"][0,:N]=torch.tensor([len(h["ids"])-1 for h in hyps],dtype=torch.int32); B["nlive"][0]=N
        _hip.check(_hip.lib().las_beam_step(_hip.p(B["lg"]),_hip.p(B["score"]),_hip.p(B["length"]),_hip.p(B["nlive"]),1,beam,V,64,t,1,_hip.p(B["par"]),_hip.p(B["tok"]),_hip.p(B["osc"]),_hip.p(B["on"]),_hip.stream()),"bs")
        n=int(B["on"][0]); par=B["par"][0,:n].tolist(); tok=B["tok"][0,:n].tolist(); osc=B["osc"][0,:n].tolist()
        exp=[(k[1],k[3]) for k in top]
        if list(zip(par,tok))!=exp:
            print("seed",c["seed"],"t",t,"N",N,"n",n,"len(top)",len(top)); print(" exp",[(k[1],k[3],float(k[0])) for k in top]); print(" got",list(zip(par,tok,osc))); bad=True; break
        nxt=[]
        for (nrm,i,l,v,s) in top:
            h=dict(ids=hyps[i]["ids"]+[v],lp=s,att=al[i],st=tuple(new[q][i:i+1] for q in range(2)))
            (sel if v==2 else nxt).append(h)
        hyps=nxt;t+=1
    print("seed",c["seed"],"ok" if not bad else "DIVERGED")
