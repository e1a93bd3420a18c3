"""Host CPU seconds the bench process burns per wall second (in-flight proofs spin or block in hipStreamSynchronize?)"""
import os, sys, subprocess, time, resource
t0 = time.perf_counter()
r = subprocess.run([sys.executable, "bench.py", "--steps", "200", "--warmup", "8", "--timed-only"] + sys.argv[1:], capture_output=True, text=True)
wall = time.perf_counter() - t0
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print(r.stdout.strip().splitlines()[-1][:200])
print("wall %.2f s, user %.2f s, sys %.2f s -> %.2f CPUs busy on average (includes start-up: import torch, circuit creation)" % (wall, ru.ru_utime, ru.ru_stime, (ru.ru_utime + ru.ru_stime) / wall))
