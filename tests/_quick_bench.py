import sys, time, numpy as np
sys.path.insert(0, '.')
import lbm_amd
print("copy GB/s", lbm_amd.copy_bandwidth_gbps(1<<30, 20))
for (nx, ny, steps) in [(1024,1024,2000),(8192,8192,200)]:
    ob = np.zeros((ny,nx), np.int32); ob[0,:]=ob[-1,:]=1; ob[:,0]=ob[:,-1]=1
    p = lbm_amd.make_params(nx, ny, steps+50, obstacles=ob)
    with lbm_amd.LBM(p, ob) as sim:
        sim.run(50)
        for nt in (0,1):
            sim.set_option("nt_stores", nt)
            for gb in (0, 1024, 2048, 8192):
                sim.set_option("grid_blocks", gb)
                sim.upload(None); sim.run(20)
                ms = sim.run_timed(steps//4)
                mlups = nx*ny*(steps//4)/(ms*1e-3)/1e6
                print(nx, ny, "nt",nt,"blocks", sim.get_option("grid_blocks"), "ms/step %.4f"%(ms/(steps//4)), "MLUPS %.0f"%mlups, "GB/s %.0f"%(mlups*72e-3), flush=True)
