import cProfile, pstats, sys, runpy
sys.argv = ["demo_synthetic_env.py", "--steps", "1500"]
pr = cProfile.Profile(); pr.enable()
try:
    runpy.run_path("examples/demo_synthetic_env.py", run_name="__main__")
finally:
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
