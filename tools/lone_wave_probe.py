import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from fractalshark_amd import GPURenderer, inputs, LAV2_PO, LAV2_FULL, PARITY_CPU, T_HDR32
for (w,h) in ((64,36),(256,144),(640,360)):
    v=inputs.View.builtin(5,w,h); ob=inputs.Orbit(v); la=inputs.LATable(ob)
    r=GPURenderer(0); r.InitializeMemory(w,h,1,None,0,0,0,False); r.InitializePerturb(1,ob,0,None,la)
    co=[(float(c["m"]),int(c["e"])) for c in v.coords_perturb_hdr32(ob)]
    for mode,name in ((LAV2_PO,'PO'),(LAV2_FULL,'FULL')):
        for lit in (False,):
            r.enable_step_count(True)
            r.RenderPerturbLAv2(None,None,None,*co,v.num_iterations,T=T_HDR32,Mode=mode,parity=PARITY_CPU); r.SyncComputeStream()
            st=r.read_step_count(); r.enable_step_count(False)
            r.RenderPerturbLAv2(None,None,None,*co,v.num_iterations,T=T_HDR32,Mode=mode,parity=PARITY_CPU); r.SyncComputeStream()
            ms=r.last_kernel_ms(); out=r.new_iter_buffer(); r.RenderCurrent(v.num_iterations,out); r.SyncComputeStream()
            mx=int(out.max())
            print(w,h,name,'ms %.2f'%ms,'max iter',mx,'steps',st['perturb_steps'],'lane util %.3f'%(st['perturb_steps']/max(1,st['lane_slots'])),'ns per max-iter step %.1f'%(ms*1e6/mx))
    r.close()
