#!/usr/bin/env python3
"""CPU-restriction sweep of the host side of the pipeline (VERDICT r3, next #1b).

Runs `bench.py --host-only --steps 300` once per CPU budget, each in a child process whose affinity mask bench.py cuts
down to its first N usable CPUs before anything touches the GPU (--cpus N), and collects the lines into one JSON file:
what one rank keeps of its frame rate when 8 ranks share a host (a GPU box gives one GPU's job 16 hardware threads; 8
ranks on a 128-core host would have 16-32 each, a busier host fewer).

    python tools/host_scaling.py gpurun_out/host_scaling.json [--cpus 0,16,8,4,2] [--steps 300] [--repeat 2] [-- extra bench.py args]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        extra = argv[argv.index("--") + 1:]
        argv = argv[:argv.index("--")]
    out_path = argv[0]
    cpus = [0, 16, 8, 4, 2]
    steps = 300
    for i, a in enumerate(argv):
        if a == "--cpus":
            cpus = [int(v) for v in argv[i + 1].split(",")]
        if a == "--steps":
            steps = int(argv[i + 1])
    repeat = 2
    for i, a in enumerate(argv):
        if a == "--repeat":
            repeat = int(argv[i + 1])
    points = []
    # every budget `repeat` times, the passes interleaved (a box shared with other tenants has bursts that last seconds: one
    # run per point put a 15 % dip on whichever point it hit); a point reports every run and the best of them
    runs = {n: [] for n in cpus}
    for rep in range(repeat):
        for n in cpus:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--host-only", "--steps", str(steps), "--warmup", "20",
                   "--cpus", str(n)] + extra
            print("host_scaling:", " ".join(cmd), flush=True)
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = None
            for ln in r.stdout.splitlines():
                if ln.startswith("{"):
                    line = json.loads(ln)
            if r.returncode != 0 or line is None:
                print("  failed:", r.stderr[-500:], flush=True)
                continue
            runs[n].append(line)
            print("  cpus=%s usable=%d steady=%.0f fps host=%s" % (n, line["host"]["usable_cpus"], line["steady_state_fps"],
                                                                   json.dumps(line["host"]["per_batch_us"])), flush=True)
    for n in cpus:
        if not runs[n]:
            points.append({"cpus_requested": n, "error": "no run completed"})
            continue
        best = max(runs[n], key=lambda ln: ln["steady_state_fps"])
        points.append({"cpus_requested": n, "usable_cpus": best["host"]["usable_cpus"], "value_fps": best["value"],
                       "steady_state_fps": best["steady_state_fps"], "all_runs_steady_state_fps": [ln["steady_state_fps"] for ln in runs[n]],
                       "host": best["host"]})
    base = next((p for p in points if p.get("cpus_requested") == 0 and "steady_state_fps" in p), None)
    for p in points:
        if base and "steady_state_fps" in p:
            p["frac_of_unconstrained"] = round(p["steady_state_fps"] / base["steady_state_fps"], 4)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump({"what": "bench.py --host-only --steps %d, one child process per CPU budget (affinity cut before the GPU is touched)" % steps,
                   "extra_args": extra, "points": points}, f, indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
