import os, sys
sys.path.insert(0, os.getcwd())
import torch, area_average_interpolation_amd as aai
W=H=8192; B=4
rq=aai.make_request(W,H,4,1,((W-1)/2,(H-1)/2),0.0); rc,msg,lay=aai.query(rq); dW,dH=lay.dst_width,lay.dst_height
aai.set_device(0)
src=torch.empty((B,H,W),dtype=torch.float32,device="cuda"); dst=torch.empty((B,dH,dW),dtype=torch.float32,device="cuda")
st=torch.cuda.current_stream().cuda_stream
for b in range(B): aai.synth_device(src[b].data_ptr(),W,H,W,b+1,st)
run=lambda: aai.resample_device(rq,src.data_ptr(),W,dst.data_ptr(),dW,st,batch=B,src_image_stride=W*H,dst_image_stride=dW*dH)
run(); torch.cuda.synchronize()
for n in (1,3,10,20,50,200):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    print("n=%3d back-to-back: %.1f us/launch"%(n,e0.elapsed_time(e1)/n*1e3))
# per-launch events like bench
evs=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a,b in evs: a.record(); run(); b.record()
torch.cuda.synchronize()
print("per-launch events:", " ".join("%.0f"%(a.elapsed_time(b)*1e3) for a,b in evs))
