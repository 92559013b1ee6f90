import sys, time, os
sys.path.insert(0, "tools")
import qz_bind as B, qz_corpus as K
plug = B.Plugin()
lane = plug.service_lane(slot=3)
blk = K.by_name("system", 131072, seed=5)
lane.run(blk, 1)
def t(): 
    t0 = time.perf_counter(); lane.run(blk, 1); return (time.perf_counter() - t0) * 1e6
warm = sorted(t() for _ in range(20))[10]
cold = []
for _ in range(8):
    time.sleep(0.06); cold.append(t())
print("resident: %.0f us per request (python lane); after 60 ms of idling (relaunch): %s" % (warm, ["%.0f" % c for c in cold]))
plug.lib.qzstd_hip_service_stop(0); lane.close()
