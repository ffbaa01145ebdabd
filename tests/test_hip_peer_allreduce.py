"""The one-shot peer-memory all-reduce (csrc/dpll_allreduce.hpp): several ranks sharing ONE GPU exchange
IPC handles and reduce; results must equal the exact sum, be identical on every rank, and keep working under
hipGraph replay (the call counter lives in device memory).  Needs the MI355X: `pytest -m gpu`."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ASSET_DIR, GOLDEN_DIR, REPO

pytestmark = pytest.mark.gpu


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.distributed import GradientAllReduce, PeerAllReduce, shard_bounds
    peer = PeerAllReduce()
    assert peer.self_test()
    gen = torch.Generator().manual_seed(100 + rank)
    # eager calls, float and double, different lengths
    for step in range(50):
        for dtype, n in ((torch.float32, 16), (torch.float64, 30), (torch.float32, 256), (torch.float64, 1)):
            mine = torch.rand(n, generator=gen, dtype=torch.float64)
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            t = mine.to(dtype).cuda()
            peer.all_reduce(t)
            expect = sum(e.to(dtype).double() for e in everyone)
            assert (t.cpu().double() - expect).abs().max() <= (1e-6 if dtype == torch.float32 else 1e-15) * world
    assert peer.healthy()
    # graph replay advances the call counter on the device
    t = torch.full((16,), float(rank + 1), device='cuda')
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        peer.all_reduce(t)
    torch.cuda.current_stream().wait_stream(side)
    dist.barrier()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        peer.all_reduce(t)
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    assert peer.healthy()
    total = world * (world + 1) / 2
    # t was reduced once eagerly (-> total), then 5 replays each multiply the common value by `world`
    assert torch.allclose(t, torch.full((16,), total * world**5, device='cuda')), (t[0].item(), total * world**5)
    peer.close()

    # end to end: sharded fused loss + peer all-reduce == single-process full batch
    g = np.load(os.path.join(GOLDEN_DIR, 'cube_box_literal.npz'))
    system = MultibodyLearnableSystem({'cube': os.path.join(ASSET_DIR, 'cube.urdf')}, float(g['dt']), dtype=torch.float64,
                                      device='cuda:0')
    batch = 301
    x = torch.tensor(g['x'][:batch], device='cuda:0')
    xp = torch.tensor(g['x_plus'][:batch], device='cuda:0')
    full = system.contactnets_loss_and_grad(x, xp).clone()
    full_grad = system.grad_buffer().clone()
    reducer = GradientAllReduce(system, global_batch=batch, transport='peer')  # ('auto' is RCCL unless DPLL_PEER_EXCHANGE=1)
    assert reducer.transport == 'peer'
    lo, hi = shard_bounds(batch, rank, world)
    system.contactnets_loss_and_grad(x[lo:hi], xp[lo:hi])
    reduced = reducer.all_reduce_mean()
    assert (reduced - full_grad).abs().max() < 1e-15
    assert reducer.fused and system._fused_ar is not None  # the exchange ran inside the loss launch's finalize kernel
    fused_result = reduced.clone()
    # the fused step under hipGraph replay
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        system.contactnets_loss_and_grad(x[lo:hi], xp[lo:hi])
        reducer.all_reduce_mean()
    torch.cuda.current_stream().wait_stream(side)
    dist.barrier()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        system.contactnets_loss_and_grad(x[lo:hi], xp[lo:hi])
        reducer.all_reduce_mean()
    for _ in range(3):
        system.grad_buffer().zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(system.grad_buffer(), fused_result)
    assert reducer.peer.healthy()
    # a separate exchange kernel after the launch gives bitwise the same row
    unfused = GradientAllReduce(system, global_batch=batch, transport='peer', fuse=False)
    assert unfused.transport == 'peer' and not unfused.fused and system._fused_ar is None
    system.contactnets_loss_and_grad(x[lo:hi], xp[lo:hi])
    assert torch.equal(unfused.all_reduce_mean(), fused_result)
    # a tail batch smaller than the world: ranks with an empty shard launch a zero row and still take part (fused exchange)
    tail = GradientAllReduce(system, global_batch=1, transport='peer')
    lo1, hi1 = shard_bounds(1, rank, world)
    system.contactnets_loss_and_grad(x[lo1:hi1], xp[lo1:hi1])
    tail_reduced = tail.all_reduce_mean().clone()
    tail.check_healthy()
    single = MultibodyLearnableSystem({'cube': os.path.join(ASSET_DIR, 'cube.urdf')}, float(g['dt']), dtype=torch.float64,
                                      device='cuda:0')
    single.contactnets_loss_and_grad(x[:1], xp[:1])
    assert (tail_reduced - single.grad_buffer()).abs().max() < 1e-15
    # links welded on with rows of their own (tests/test_welded_links.py): the rows' gradient is not a view of the exchanged buffer,
    # it is chained again from the summed theta block (system.after_grad_reduce) -- sharded + exchange == the full batch
    gw = np.load(os.path.join(GOLDEN_DIR, 'welded_arm_literal.npz'))
    welded = MultibodyLearnableSystem({'welded_arm': os.path.join(ASSET_DIR, 'welded_arm.urdf')}, float(gw['dt']), dtype=torch.float64,
                                      device='cuda:0')
    xw, xpw = torch.tensor(gw['x'], device='cuda:0'), torch.tensor(gw['x_plus'], device='cuda:0')
    theta = welded.multibody_terms.lagrangian_terms.inertial_parameters
    welded.contactnets_loss_and_grad(xw, xpw)
    whole = theta.grad.clone()
    assert whole.shape == (5, 10)
    wred = GradientAllReduce(welded, global_batch=xw.shape[0], transport='peer')
    lo, hi = shard_bounds(xw.shape[0], rank, world)
    welded.contactnets_loss_and_grad(xw[lo:hi], xpw[lo:hi])
    wred.all_reduce_mean()
    wred.check_healthy()
    assert (theta.grad - whole).abs().max() <= 1e-13 * whole.abs().max()
    np.save(os.path.join(out_dir, f'rank{rank}.npy'), reduced.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_peer_allreduce_on_one_gpu(tmp_path, world):
    port = 29700 + os.getpid() % 1000 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f'rank{r}.npy') for r in range(world)]
    for other in outs[1:]:
        assert np.array_equal(outs[0], other)  # bitwise identical on every rank


def _xgmi_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    """one process per GPU, RCCL process group: the peer-memory exchange across REAL xGMI peers against RCCL's all-reduce"""
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    device = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.distributed import GradientAllReduce, shard_bounds
    g = np.load(os.path.join(GOLDEN_DIR, 'cube_box_literal.npz'))
    batch = 301
    x = torch.tensor(g['x'][:batch], device=device)
    xp = torch.tensor(g['x_plus'][:batch], device=device)
    lo, hi = shard_bounds(batch, rank, world)
    results = {}
    for transport in ('collective', 'auto', 'peer'):
        system = MultibodyLearnableSystem({'cube': os.path.join(ASSET_DIR, 'cube.urdf')}, float(g['dt']), dtype=torch.float64,
                                          device=str(device))
        reducer = GradientAllReduce(system, global_batch=batch, transport=transport)
        system.contactnets_loss_and_grad(x[lo:hi], xp[lo:hi])
        results[transport] = (reducer.all_reduce_mean().clone(), reducer.transport)
        reducer.check_healthy()
    full = MultibodyLearnableSystem({'cube': os.path.join(ASSET_DIR, 'cube.urdf')}, float(g['dt']), dtype=torch.float64,
                                    device=str(device))
    full.contactnets_loss_and_grad(x, xp)
    for transport, (reduced, used) in results.items():
        assert (reduced - full.grad_buffer()).abs().max() < 1e-14, (transport, used)
    assert results['auto'][1] == 'collective' or os.environ.get('DPLL_PEER_EXCHANGE') == '1'  # RCCL unless opted in
    np.save(os.path.join(out_dir, f'rank{rank}.npy'), results['peer'][0].cpu().numpy())
    with open(os.path.join(out_dir, f'transport{rank}.txt'), 'w') as handle:
        handle.write(results['peer'][1])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs of one node (RCCL over xGMI); the GPU test box has one')
def test_gradient_exchange_across_two_gpus(tmp_path):
    """N > 1 as the driver launches it (one process per GPU, backend nccl = RCCL): sharded loss + one exchange equals the
    full batch with RCCL's all-reduce and with the one-shot peer-memory kernel, identical on both ranks."""
    port = 29800 + os.getpid() % 1000
    mp.spawn(_xgmi_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / 'rank0.npy'), np.load(tmp_path / 'rank1.npy')
    assert np.array_equal(a, b)
