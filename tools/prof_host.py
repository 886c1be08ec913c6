import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-graph", "--no-cpu-baseline", "--no-secondary"]
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
