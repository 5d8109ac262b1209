"""CPU: C-ABI library loads and exports every declared symbol; host-side mirror (state_dict keys, CLI, dataset protocol);
the product path refuses to run without a GPU; data-parallel plumbing over gloo (world_size 2)."""
import json
import os
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_library_exports_every_declared_symbol():
    from tcct_amd._lib import lib, parse_header
    protos = parse_header()
    assert len(protos) >= 60
    dll = lib.load()
    for name in protos:
        assert hasattr(dll, name), name
    assert dll.tcct_version() >= 100
    # every prototype in the header is extern "C" int / int64_t / const char*; argument parsing keeps arity
    assert len(protos['tcct_conv2d_fwd'][1]) == 18
    assert len(protos['tcct_clip_adamw'][1]) == 16


def test_build_entry_point_is_idempotent():
    import __graft_entry__ as g
    g.build()


def test_state_dict_keys_match_reference():
    from tcct_amd.nets import stc_tt, RegNet, tcct
    assert tcct is stc_tt
    m = RegNet(stc_tt(5), con='cos', out_channels=5)
    ref = {k: tuple(s) for k, s in json.load(open(os.path.join(HERE, 'golden', 'state_dict_keys.json')))}
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == ref
    assert sum(p.numel() for p in m.base.parameters()) == 992028
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 992187
    assert m.__name__ == 'stctt' and hasattr(m.base, 'base_cnn') and hasattr(m.base, 'base_vit')
    # shared ConvPosEnc aliases (reference tcct.py:491-505)
    st = m.base.base_vit.mhca_stages[0].mhca_blks[0]
    assert st.cpe is st.MHCA_layers[0].cpe
    # loading a reference-keyed checkpoint works with strict=True
    m.load_state_dict({k: torch.zeros(s) if 'num_batches' not in k else torch.zeros((), dtype=torch.int64) for k, s in ref.items()}, strict=True)


def test_product_path_has_no_cpu_fallback():
    from tcct_amd.nets import stc_tt
    from tcct_amd._lib import TcctError
    m = stc_tt(5)
    with pytest.raises(TcctError):
        m(torch.zeros(1, 3, 32, 32))
    from tcct_amd import ops
    with pytest.raises(TcctError):
        ops.act(torch.zeros(4, 4), 'gelu')
    # and the package never imports the oracle
    src = subprocess.run(['grep', '-rl', 'tcct_oracle', os.path.join(ROOT, 'tcct_amd')], capture_output=True, text=True).stdout
    assert src.strip() == ''


def test_cli_flags_match_reference_surface():
    from tcct_amd.kite.main import parse_args, build_parser, str2bool
    a = parse_args([])
    for f in ('db lr wd inc gpu los net pth bs epochs root resume reg coff_reg epl coff_epl udh coff_udh type_udh ds coff_ds '
              'pl bug').split():
        assert hasattr(a, f), f
    assert (a.coff_reg, a.coff_udh, a.coff_ds, a.type_udh, a.net) == (0.1, 1, 1, 'cos', 'stc_tt') and a.graph is False
    b = parse_args(['--los=di+reg+fpl', '--bs=8', '--pl=true'])
    assert b.los == 'di' and b.reg and b.udh and b.pl and b.bs == 8
    assert str2bool('Yes') and not str2bool('0')
    import argparse
    with pytest.raises(argparse.ArgumentTypeError):
        str2bool('maybe')
    assert isinstance(build_parser(), argparse.ArgumentParser)


def test_synthetic_dataset_protocol():
    from tcct_amd.data import EyeSetGenerator
    ds = EyeSetGenerator('synth', height=64, width=100, device='cpu', n_train=4)
    assert ds.out_channels == 5
    batches = list(ds.trainSet(bs=2))
    assert len(batches) == 2
    img, lab, tag, _ = ds.parse(batches[0])
    assert img.shape == (2, 1, 64, 112) and lab.shape == (2, 64, 112) and lab.dtype == torch.int64
    assert 0 <= img.min() and img.max() < 1 and len(tag) == 2
    assert (img[..., 100:] == 0).all() and (lab[..., 100:] == 0).all()
    for c in range(5):
        assert (batches[0]['lab'] == c).sum() > 32 * 2
    # deterministic in the seed
    assert torch.equal(ds.make_batch(2, 7)['lab'], ds.make_batch(2, 7)['lab'])


def test_lazy_mask_and_loss_factory():
    from tcct_amd.kite.losses import get_loss, MaskOneHot
    from tcct_amd._lib import TcctError
    crit = get_loss('di')
    assert crit.__class__.__name__ == 'MultiLoss'
    with pytest.raises(TcctError):
        get_loss('ce')
    m = MaskOneHot(torch.tensor([[[0, 1], [2, 4]]], dtype=torch.uint8), 5)
    d = m.dense()
    assert d.shape == (1, 5, 2, 2) and d.sum().item() == 4 and d[0, 4, 1, 1] == 1


DDP_SCRIPT = r'''
import os, sys, torch
sys.path.insert(0, %r)
from tcct_amd import dist as tdist
world, rank, local = tdist.init(backend='gloo')
assert world == 2
# shard a global batch of 8 and all-reduce a flat "gradient" buffer; every rank must end with the sum over ranks
sl = tdist.shard_batch(8, world, rank)
assert (sl.start, sl.stop) == (rank * 4, rank * 4 + 4)
g = torch.Generator().manual_seed(123)
per_sample = torch.randn(8, 1000, generator=g)            # identical on both ranks
flat = per_sample[sl].sum(0).clone()
tdist.allreduce_sum_(flat)
assert torch.allclose(flat, per_sample.sum(0), atol=1e-5)
# parameters broadcast from rank 0
lin = torch.nn.Linear(4, 4)
with torch.no_grad():
    lin.weight.fill_(float(rank + 1))
tdist.broadcast_params_(lin)
assert lin.weight.eq(1.0).all()
t = tdist.max_over_ranks(float(rank), torch.device('cpu'))
assert t == 1.0
assert tdist.world_rank() == (2, rank) and tdist.gather_strings('r%%d' %% rank) == ['r0', 'r1']
# bucketed all-reduce: readiness marks may arrive out of order, launches never do (0, 1 during "backward", 2 at step())
gb = tdist.GradBuckets()
flat = per_sample[sl].sum(0).clone()
gb.bind(flat, [300, 500, 200])
gb.begin_step(armed=True)
gb.on_mark('deep')
assert gb.launch_log == [] and not gb.is_launched(0)
gb.on_mark('dec')
assert gb.launch_log == [(0, 'backward'), (1, 'backward')] and gb.is_launched(799) and not gb.is_launched(800)
gb.finish()
assert gb.launch_log[-1] == (2, 'step') and torch.allclose(flat, per_sample.sum(0), atol=1e-5)
gb.begin_step(armed=False)                               # not armed (first step / plain backward): everything leaves in finish()
flat.copy_(per_sample[sl].sum(0))
gb.on_mark('dec')
gb.finish()
assert gb.launch_log == [(0, 'step'), (1, 'step'), (2, 'step')] and torch.allclose(flat, per_sample.sum(0), atol=1e-5)
# fallback: a gradient of bucket 0 arrived through autograd (p.grad is a tensor that is NOT its slot of the flat buffer) -> step() will copy it
# in later, an early launch would send stale data: the early launches are called off, everything leaves in finish(), sums still right
flat.copy_(per_sample[sl].sum(0))
slot_p = torch.nn.Parameter(torch.zeros(300)); slot_p._grad_slot = flat[:300]; slot_p.grad = None           # written in place by a kernel
auto_p = torch.nn.Parameter(torch.zeros(500)); auto_p._grad_slot = flat[300:800]; auto_p.grad = torch.ones(500)      # delivered by autograd
tail_p = torch.nn.Parameter(torch.zeros(200)); tail_p._grad_slot = flat[800:]
gb2 = tdist.GradBuckets()
gb2.bind(flat, [300, 500, 200], [slot_p, auto_p, tail_p])
gb2.begin_step(armed=True)
gb2.on_mark('dec'); gb2.on_mark('deep')
assert gb2.launch_log == [(0, 'backward')] and gb2.fallbacks == 1 and not gb2.armed, (gb2.launch_log, gb2.fallbacks)
gb2.finish()
assert gb2.launch_log == [(0, 'backward'), (1, 'step'), (2, 'step')] and torch.allclose(flat, per_sample.sum(0), atol=1e-5)
gb2.begin_step(armed=True)                 # the same with the offending gradient in bucket 0: nothing leaves early
flat.copy_(per_sample[sl].sum(0))
slot_p.grad, auto_p.grad = torch.ones(300), None
gb2.on_mark('dec'); gb2.on_mark('deep')
assert gb2.launch_log == [] and gb2.fallbacks == 2
gb2.finish()
assert [w for _, w in gb2.launch_log] == ['step'] * 3 and torch.allclose(flat, per_sample.sum(0), atol=1e-5)
# BatchNorm buffers averaged before validation: every rank evaluates the same model
bn = torch.nn.BatchNorm2d(3)
with torch.no_grad():
    bn.running_mean.fill_(float(rank)); bn.running_var.fill_(1.0 + 2.0 * rank)
tdist.average_buffers_(bn)
assert torch.allclose(bn.running_mean, torch.full((3,), 0.5)) and torch.allclose(bn.running_var, torch.full((3,), 2.0)) and bn.num_batches_tracked.item() == 0
# per-rank noise streams (SURVEY 8(e): seed = base + rank): the two replicas must draw DIFFERENT DropPath masks
import types
from tcct_amd.kite.loop_seg import KiteSeg
from tcct_amd.kite.loopback import setup_seed
from tcct_amd.nets.tcct import MPViT
seed = KiteSeg.epoch_seed(types.SimpleNamespace(rank=rank), 3)
assert seed == 3 * 311 + 2023 + rank
setup_seed(seed)
vit = MPViT().train()
masks = torch.stack([torch.stack(pair) for pair in vit._dp_scales(64, torch.device('cpu')) if pair is not None]).reshape(-1)
import torch.distributed as tdd
both = [torch.zeros_like(masks) for _ in range(2)]
tdd.all_gather(both, masks)
assert not torch.equal(both[0], both[1]), 'both ranks drew the same DropPath masks'
tdist.barrier()
os.write(1, ('rank %%d ok\n' %% rank).encode())      # one atomic write: the two ranks share stdout
'''


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        return s_.getsockname()[1]


def test_data_parallel_plumbing_gloo_world2(tmp_path):
    script = tmp_path / 'ddp.py'
    script.write_text(DDP_SCRIPT % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'rank 0 ok' in r.stdout and 'rank 1 ok' in r.stdout


def test_reference_checkpoint_layouts():
    """the two checkpoint layouts the reference ships (tests/golden/ckpt_*.npz = real trained weights of task1/onnx/*.pt, bf16-rounded):
    layout / class-count detection, strict=False key report identical to the reference's own for the current layout, loud errors"""
    import numpy as np
    from tcct_amd import checkpoint as C
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd._lib import TcctError
    duke = C.read_checkpoint(os.path.join(HERE, 'golden', 'ckpt_duke.npz'))
    goals = C.read_checkpoint(os.path.join(HERE, 'golden', 'ckpt_goals_legacy.npz'))
    assert C.describe(duke) == dict(n_class=9, legacy_heads=False) and C.describe(goals) == dict(n_class=5, legacy_heads=True)
    z = np.load(os.path.join(HERE, 'golden', 'ckpt_duke.npz'))
    net = RegNet(stc_tt(9), out_channels=9)
    missing, unexpected = C.load_reference_checkpoint(net, duke)
    assert missing == sorted(str(k) for k in z['missing']) and unexpected == sorted(str(k) for k in z['unexpected'])
    assert torch.equal(net.base.aux0.weight.detach(), duke['base.aux0.weight']) and net.base.base_cnn.cnn[1].num_batches_tracked.item() > 0
    leg = RegNet(stc_tt(5, legacy_heads=True), out_channels=5)
    assert not any(k.startswith('base.t32') for k in leg.state_dict())
    m2, u2 = C.load_reference_checkpoint(leg, goals)
    assert all(k.startswith(('lap_map', 'tau')) for k in m2) and all(k.startswith(('aug.', 'lap_reg.2')) for k in u2)
    with pytest.raises(TcctError):
        C.load_reference_checkpoint(RegNet(stc_tt(5), out_channels=5), goals)           # current-layout model, legacy file
    with pytest.raises(TcctError):
        C.load_reference_checkpoint(RegNet(stc_tt(5), out_channels=5), duke)            # 9-class file, 5-class model
    with pytest.raises(TcctError):
        C.describe({'x': torch.zeros(1)})


def test_token_mixer_option_mirrors_the_reference_constructor():
    """att='pool' keeps the reference's 514 state_dict keys; att='factor' adds exactly the parameters the commented-out
    FactorAtt_ConvRelPosEnc would register (reference nets/tcct.py:289-341, 443-448): qkv (with bias, tcct.py:424) / proj per block and
    the `att.crpe` aliases of the shared ConvRelPosEnc, whose 2+3+3 head splits now divide 8 heads (tcct.py:484-488)"""
    import pytest
    from tcct_amd.nets import stc_tt
    from tcct_amd.nets.tcct import ConvRelPosEnc
    from tcct_amd.kite.main import parse_args
    base = stc_tt(5)
    fa = stc_tt(5, att='factor')
    kb, kf = set(base.state_dict()), set(fa.state_dict())
    extra = kf - kb
    assert kb <= kf and len(extra) == 4 * (4 + 6)
    assert all('.MHCA_layers.0.att.' in k for k in extra)
    for s, dim in enumerate((64, 96, 128, 160)):
        blk = f'base_vit.mhca_stages.{s}.mhca_blks.0'
        sd = fa.state_dict()
        assert tuple(sd[f'{blk}.MHCA_layers.0.att.qkv.weight'].shape) == (3 * dim, dim)
        assert tuple(sd[f'{blk}.MHCA_layers.0.att.qkv.bias'].shape) == (3 * dim,)
        Ch = dim // 8
        assert [tuple(sd[f'{blk}.crpe.conv_list.{i}.weight'].shape) for i in range(3)] == [(2 * Ch, 1, 3, 3), (3 * Ch, 1, 5, 5), (3 * Ch, 1, 7, 7)]
        # shared module: the alias keys are the same storage
        assert sd[f'{blk}.MHCA_layers.0.att.crpe.conv_list.0.weight'].data_ptr() == sd[f'{blk}.crpe.conv_list.0.weight'].data_ptr()
    with pytest.raises(ValueError):
        stc_tt(5, att='bogus')
    with pytest.raises(ValueError):
        ConvRelPosEnc(Ch=8, h=8, window='3')                    # reference tcct.py:245
    assert ConvRelPosEnc(Ch=8, h=8, window=3).channel_splits == [64]
    assert parse_args(['--att=factor']).att == 'factor' and parse_args([]).att == 'pool'


def test_gradient_buckets_follow_the_backward_order():
    """static bucket of every trained parameter (tcct_amd.dist.bucket_of): decoder / fusion first, deep encoder levels second,
    level 0 + stems + the loss-side modules last; the three buckets cover the 802 298 trained elements of --reg=true"""
    import json
    from tcct_amd.dist import bucket_of, N_BUCKETS
    keys = json.load(open(os.path.join(HERE, 'golden', 'state_dict_keys.json')))
    names = [k for k, _ in keys]
    assert N_BUCKETS == 3
    assert bucket_of('base.dec1.prep.0.weight') == 0 and bucket_of('base.tran_vit2.1.bias') == 0 and bucket_of('base.aux4.weight') == 0
    assert bucket_of('base.base_cnn.path_estan.3.block5.0.weight') == 1 and bucket_of('base.base_vit.mhca_stages.2.aggregate.bn.weight') == 1
    assert bucket_of('base.base_vit.patch_embed_stages.1.patch_embeds.0.patch_conv.pwconv.weight') == 1
    assert bucket_of('base.base_cnn.path_estan.0.block12.0.weight') == 2 and bucket_of('base.base_cnn.cnn.0.weight') == 2
    assert bucket_of('base.base_vit.stem.0.conv.weight') == 2 and bucket_of('base.base_vit.mhca_stages.0.InvRes.conv1.conv.weight') == 2
    assert bucket_of('lap_reg.0.weight') == 2 and bucket_of('lap_map.1.weight') == 2
    assert {bucket_of(n) for n in names} == {0, 1, 2}


def test_bench_launcher_starts_the_ranks_as_children_and_fails_loudly():
    """`python bench.py --gpus 2` WITHOUT a torchrun environment must start two ranks itself (child process, torch.distributed.run) and
    relay their exit status: here (no GPU) the ranks refuse to run, and the launcher must report that instead of printing a 1-rank
    line.  With WORLD_SIZE set the process is a rank: a mismatch with --gpus is an error."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert 'torch.distributed.run' in r.stderr and '--nproc-per-node=2' in r.stderr
    assert r.stderr.count('bench.py needs an MI355X') >= 1 and 'child exited with code' in r.stderr
    assert '"metric"' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'), timeout=300)
    assert r.returncode != 0 and '--gpus=4 but WORLD_SIZE=2' in r.stderr and '"metric"' not in r.stdout


def test_bench_eight_ranks_through_the_launcher_stub_trainer():
    """`python bench.py --gpus 8` end to end through the launcher -- child torch.distributed.run, 8 ranks, rendezvous on 127.0.0.1, process group,
    barriers, max-over-ranks timing, one all-reduce of the real flat-gradient size per step, the rank-evidence all-reduce, rank 0 alone printing, all
    ranks leaving together -- with the GPU work replaced by bench.py's stub trainer (TCCT_BENCH_STUB=1: gloo, CPU tensors).  The 5-rank rehearsal of
    round 3 found a real deadlock in this sequence; 8 ranks is what the driver's SCALE run starts."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(TCCT_BENCH_STUB='1', OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                        # exactly ONE line on stdout
    got = json.loads(lines[0])
    assert got['n_gpus'] == 8 and got['config']['ranks'] == 8 and got['config']['allreduce_ranks_seen'] == 8
    assert got['config']['global_batch'] == 64 and got['config']['parallelism'] == 'dp8' and got['scaling'] == 'weak'
    assert got['config']['backend'] == 'gloo' and len(got['config']['rank_devices']) == 8
    assert got['metric'].startswith('STUB') and got['data'].startswith('stub') and 'roofline' not in got and 'cpu_baseline' not in got
    assert got['steps'] == 3 and got['value'] > 0


def test_ranks_slice_the_same_shuffled_global_batches_with_a_single_process_loader():
    """with `shuffle=True, num_workers=0` RandomSampler draws its seed lazily at the first next(); KiteSeg must fix the loader order
    under the COMMON epoch seed before it switches to the per-rank noise seed, or every rank slices a different global batch"""
    import types
    from torch.utils.data import DataLoader, TensorDataset
    from tcct_amd.kite.loop_seg import KiteSeg

    class DS:
        def trainSet(self, bs):
            return DataLoader(TensorDataset(torch.arange(64)), batch_size=bs, shuffle=True, num_workers=0)
    orders, noise = [], []
    for rank in (0, 1, 2):
        ns = types.SimpleNamespace(dataset=DS(), args=types.SimpleNamespace(bs=4), world=4, rank=rank)
        ns.epoch_seed = lambda e, ns=ns: KiteSeg.epoch_seed(ns, e)
        orders.append(torch.cat([b[0] for b in KiteSeg._global_batches(ns, 3)]))
        noise.append(torch.rand(4))
    assert sorted(orders[0].tolist()) == list(range(64)) and orders[0].tolist() != list(range(64))        # a permutation, shuffled
    assert torch.equal(orders[0], orders[1]) and torch.equal(orders[0], orders[2])                       # the same on every rank
    assert not torch.equal(noise[0], noise[1]) and not torch.equal(noise[1], noise[2])                   # ... while the noise streams differ
