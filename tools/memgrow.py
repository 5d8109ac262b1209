import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
a = argparse.Namespace(los=sys.argv[1] if len(sys.argv) > 1 else 'di', bs=8, height=800, width=1100, dtype='bf16', att='pool')
k, ds, _ = bench.build_trainer(a, 1)
k.model.train()
img, lab, _, _ = ds.parse(ds.make_batch(8, seed=5))
t0 = time.time()
for it in range(int(os.environ.get('MEMGROW_STEPS', '401'))):
    loss = k.train_step(img, lab)
    if it % 20 == 0:
        if os.environ.get('MEMGROW_SYNC', '0') == '1':
            torch.cuda.synchronize()
        ms = torch.cuda.memory_stats()
        print(it, 'segments', ms['num_device_alloc'], 'reserved GB', round(ms['reserved_bytes.all.current'] / 2**30, 2), 'allocated GB', round(ms['allocated_bytes.all.current'] / 2**30, 2), 'retries', ms['num_alloc_retries'], round(time.time() - t0, 1), 's', flush=True)
