import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for shots in [int(a) for a in sys.argv[2:]] or [8]:
    try:
        t0 = time.time()
        wl = bench.AcousticMarmousi(torch.device("cuda:0"), 0, 1, nt=int(sys.argv[1]), shots=shots)
        torch.cuda.synchronize()
        print("shots", shots, "forward OK in", round(time.time() - t0, 3), flush=True)
        wl.step(True); torch.cuda.synchronize()
        wl._ev = []
        wl.step(True); wl.step(True); torch.cuda.synchronize()
        print("   step OK; per-step us fwd/adj:", [round(x * 1e6, 2) for x in wl.kernel_times()], flush=True)
    except Exception as e:
        print("shots", shots, "FAILED:", str(e)[:200], flush=True)
