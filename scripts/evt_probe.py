import torch, time
s = torch.cuda.current_stream()
x = torch.zeros(1024, device="cuda")
ev = torch.cuda.Event()
for name, fn in (("synchronize", lambda: ev.synchronize()), ("query", lambda: ev.query())):
    x.add_(1); ev.record(s); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    print(name, "on a complete event: %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))
# fresh event each time, complete by the time we ask
tot = 0.0
for _ in range(300):
    x.add_(1); ev.record(s); time.sleep(0.0005)
    t0 = time.perf_counter(); ev.synchronize(); tot += time.perf_counter() - t0
print("synchronize 0.5 ms after record: %.2f us" % (tot / 300 * 1e6))
tot = 0.0
for _ in range(300):
    x.add_(1); ev.record(s); time.sleep(0.0005)
    t0 = time.perf_counter(); ok = ev.query(); tot += time.perf_counter() - t0
print("query 0.5 ms after record: %.2f us (%s)" % (tot / 300 * 1e6, ok))
