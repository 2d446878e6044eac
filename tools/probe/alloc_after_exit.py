"""What does a large HBM allocation cost right after ANOTHER process released its memory?  (r06: whole-SCF wall of back-to-back runs
differed by 5-7 s with identical per-cycle times; the time sat in the first tensor-sized allocation.)
    python tools/probe/alloc_after_exit.py            # parent: dirty N GB in a child, let it exit, then time fresh allocations
"""
import subprocess, sys, time
import torch


def alloc_ms(gb, fill):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = torch.empty(int(gb * (1 << 30)) // 8, dtype=torch.float64, device='cuda')
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if fill:
        x.zero_()
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    del x
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1), round((t3 - t2) * 1e3, 1)


if len(sys.argv) > 1 and sys.argv[1] == 'child':
    gb = float(sys.argv[2])
    x = torch.empty(int(gb * (1 << 30)) // 8, dtype=torch.float64, device='cuda')
    x.fill_(1.0)
    torch.cuda.synchronize()
    print('child: dirtied %.0f GB' % gb, flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == 'measure':
    torch.zeros(1, device='cuda')
    gb = float(sys.argv[2])
    print(alloc_ms(gb, True), alloc_ms(gb, True), flush=True)
    sys.exit(0)
torch.zeros(1, device='cuda')
print('fresh process, 123 GB: (alloc, zero-fill, free) ms', alloc_ms(123, True), 'again', alloc_ms(123, True), flush=True)
torch.cuda.empty_cache()
for dirty, want, wait in ((200, 123, 0), (200, 61, 0), (200, 123, 5), (100, 200, 0)):
    subprocess.run([sys.executable, __file__, 'child', str(dirty)], check=True)
    time.sleep(wait)
    out = subprocess.run([sys.executable, __file__, 'measure', str(want)], capture_output=True, text=True)
    print('after a child that dirtied %d GB exited (+%d s): %d GB (alloc, zero-fill, free) ms, twice: %s %s'
          % (dirty, wait, want, out.stdout.strip(), ''), flush=True)
