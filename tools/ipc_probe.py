"""Can two UNRELATED processes on this box hand a device buffer to each other through a pickled IPC descriptor
(torch.multiprocessing.reductions.reduce_tensor / rebuild_cuda_tensor over hipIpcGetMemHandle)?  Role A exports 80 MB
rows and writes the pickled descriptors to a file; role B (started independently) rebuilds, copies, checks, releases.
Prints sizes and timings.  usage: ipc_probe.py  (runs both roles as subprocesses)"""
import os, pickle, subprocess, sys, time

D = '/tmp/ipc_probe'


def role_a():
    import torch
    from torch.multiprocessing.reductions import reduce_tensor
    n = 10_000_000
    rows = [torch.arange(n, dtype=torch.int64, device='cuda:0') * (j + 3) for j in range(4)]
    torch.cuda.synchronize()
    for j, r in enumerate(rows):
        t0 = time.perf_counter()
        fn, args = reduce_tensor(r)
        blob = pickle.dumps((fn, args))
        dt = time.perf_counter() - t0
        with open(f'{D}/row{j}.tmp', 'wb') as fh:
            fh.write(blob)
        os.rename(f'{D}/row{j}.tmp', f'{D}/row{j}.pkl')
        print(f'A: exported row {j}: descriptor {len(blob)} bytes in {dt*1e3:.2f} ms', flush=True)
    del rows, r                                  # the exporter keeps the storage alive until the consumer lets go
    t0 = time.time()
    while not os.path.exists(f'{D}/done') and time.time() - t0 < 60:
        time.sleep(0.01)
    torch.cuda.ipc_collect()
    print('A: memory still allocated after release: %.0f MB' % (torch.cuda.memory_allocated() / 1e6), flush=True)


def role_b():
    import torch
    torch.zeros(1, device='cuda:0')
    n = 10_000_000
    for j in range(4):
        while not os.path.exists(f'{D}/row{j}.pkl'):
            time.sleep(0.001)
        blob = open(f'{D}/row{j}.pkl', 'rb').read()
        t0 = time.perf_counter()
        fn, args = pickle.loads(blob)
        t = fn(*args)
        mine = t.clone()
        del t
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = bool((mine[:5].cpu() == torch.arange(5) * (j + 3)).all()) and int(mine[-1]) == (n - 1) * (j + 3)
        print(f'B: row {j}: rebuilt + copied 80 MB in {dt*1e3:.2f} ms, ok={ok}', flush=True)
    open(f'{D}/done', 'w').close()


if __name__ == '__main__':
    if len(sys.argv) > 1:
        {'a': role_a, 'b': role_b}[sys.argv[1]]()
    else:
        import shutil
        shutil.rmtree(D, ignore_errors=True)
        os.makedirs(D)
        print('HSA_ENABLE_IPC_MODE_LEGACY =', os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))
        pa = subprocess.Popen([sys.executable, __file__, 'a'])
        pb = subprocess.Popen([sys.executable, __file__, 'b'])
        print('exit codes', pa.wait(timeout=120), pb.wait(timeout=120))
