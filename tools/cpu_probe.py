import sys, os, time, json, torch
sys.path.insert(0, 'oracle')
import tcct_oracle as O
keys = [(k, tuple(s)) for k, s in json.load(open('tests/golden/state_dict_keys.json'))]
print('cpu_count', os.cpu_count())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' ")
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    sd = O.formula_state_dict(keys)
    names = [k for k, v in sd.items() if v.is_floating_point() and not k.endswith(('running_mean', 'running_var')) and not k.startswith('fcp.')]
    for n in names: sd[n].requires_grad_(True)
    img, lab = O.synth_batch(1, 400, 560, seed=1)
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    for it in range(2):
        t0 = time.time()
        tot, _, _, _ = O.total_loss(sd, img, oh)
        tot.backward()
        dt = time.time() - t0
        print('threads', th, 'iter', it, f'{dt:.2f}s', flush=True)
