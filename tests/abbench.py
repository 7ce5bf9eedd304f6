import os, sys, subprocess, json
# A/B of two builds of the library in one gpurun call: python tests/abbench.py <libA> <libB> [config]
cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg4"
for rep in range(2):
    for lib in sys.argv[1:3]:
        env = dict(os.environ, P3M_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--no-cpu", "--steps", "6", "--warmup", "2"], env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print(lib, round(d["ms_per_step"], 3), {k: round(v, 4) for k, v in d["roofline"]["pass_ms"].items()}, flush=True)
