"""End-to-end: the fused loss + Adam recover perturbed cube parameters on real toss data."""
import os

import numpy as np
import pytest
import torch

from conftest import ASSET_DIR, GOLDEN_DIR

pytestmark = pytest.mark.gpu


def test_training_reduces_loss_and_matches_autograd_path():
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.trainer import ContactNetsTrainer, slice_pairs
    g = np.load(os.path.join(GOLDEN_DIR, 'cube_box_4096.npz'))
    x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
    # slice rule on a fake trajectory
    traj = torch.arange(5 * 13, dtype=torch.float32).reshape(5, 13)
    a, b = slice_pairs([traj, traj])
    assert a.shape == (8, 13) and torch.equal(a[:4], traj[:-1]) and torch.equal(b[:4], traj[1:])

    def make():
        system = MultibodyLearnableSystem({'cube': os.path.join(ASSET_DIR, 'cube.urdf')}, float(g['dt']),
                                          dtype=torch.float32, device='cuda:0')
        with torch.no_grad():  # start from a wrong shape and friction
            system.multibody_terms.contact_terms.geometries[1].length_params.mul_(1.25)
            system.multibody_terms.contact_terms.friction_params[1] = 0.6
        return system

    system = make()
    trainer = ContactNetsTrainer(system, lr=1e-3, batch_size=1024)
    log = trainer.fit(x, xp, epochs=30)
    assert log.epoch_losses[-1] < 0.6 * log.epoch_losses[0]
    half = system.multibody_terms.contact_terms.geometries[1].length_params.abs().mean().item()
    assert abs(half - 0.0524) < abs(1.25 * 0.0524 - 0.0524)  # moved towards the true half length

    # the same first optimizer step through the reference-style autograd call gives the same parameters
    s1, s2 = make(), make()
    o1 = torch.optim.Adam(s1.parameters(), lr=1e-3)
    o2 = torch.optim.Adam(s2.parameters(), lr=1e-3)
    s1.contactnets_loss_and_grad(x, xp)
    o1.step()
    o2.zero_grad()
    s2.contactnets_loss(x, torch.zeros((x.shape[0], 0), device='cuda:0'), xp).mean().backward()
    o2.step()
    for (name, p1), (_, p2) in zip(s1.named_parameters(), s2.named_parameters()):
        assert torch.allclose(p1, p2, rtol=1e-5, atol=1e-7), name


def test_prediction_loss_training_through_rollouts():
    """MultibodyLosses.PREDICTION_LOSS (drake_experiment.py:47-52, experiment.py:292-320): multi-step rollouts
    of the model itself give the targets, a model started from a wrong friction coefficient is trained through
    the rollouts (parameter gradient + state adjoint of every step) and moves back towards the truth."""
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.trainer import ContactNetsTrainer, prediction_loss, slice_windows
    g = np.load(os.path.join(GOLDEN_DIR, 'cube_box_4096.npz'))
    dt = float(g['dt'])
    urdf = {'cube': os.path.join(ASSET_DIR, 'cube.urdf')}
    truth = MultibodyLearnableSystem(urdf, dt, dtype=torch.float64, device='cuda:0')
    x0 = torch.tensor(g['x'][:256], dtype=torch.float64, device='cuda:0')
    with torch.no_grad():
        trajectories, _ = truth.simulate(x0.unsqueeze(-2), torch.zeros((256, 1), device='cuda:0'), 8)
    past, future = slice_windows(list(trajectories), 3)
    assert past.shape == (256 * 6, 1, 13) and future.shape == (256 * 6, 3, 13)
    assert torch.equal(past[1, 0], trajectories[0, 1]) and torch.equal(future[1], trajectories[0, 2:5])
    with torch.no_grad():
        assert prediction_loss(truth, past, future).item() < 1e-20  # the generating model predicts its own data
    model = MultibodyLearnableSystem(urdf, dt, dtype=torch.float64, device='cuda:0')
    with torch.no_grad():
        model.multibody_terms.contact_terms.friction_params[1] = 0.45
    # prediction losses through stiff contact are badly conditioned in the inertial and shape parameters (the
    # reason ContactNets exists); the friction coefficient alone is a well-posed one-dimensional check
    model.multibody_terms.lagrangian_terms.inertial_parameters.requires_grad_(False)
    model.multibody_terms.contact_terms.geometries[1].length_params.requires_grad_(False)
    first = prediction_loss(model, past, future)
    first.backward()
    grad_mu = model.multibody_terms.contact_terms.friction_params.grad[1].item()
    assert grad_mu > 0  # too much friction: the loss falls when it is reduced
    model.zero_grad()
    trainer = ContactNetsTrainer(model, lr=1e-2, batch_size=512, loss='prediction')
    log = trainer.fit(past, future, epochs=10)
    mu = model.multibody_terms.contact_terms.friction_params[1].abs().item()
    with torch.no_grad():
        last = prediction_loss(model, past, future).item()
    assert last < 0.7 * first.item(), (first.item(), last, log.epoch_losses, mu)
    assert abs(mu - 0.15) < abs(0.45 - 0.15), mu


def test_graph_replayed_training_step_equals_eager():
    """use_graph=True: parameter packing + fused loss/gradients + Adam captured once and replayed per batch gives
    the same parameters as the eager loop; every full batch is one graph replay (counted, not timed)."""
    import time
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.trainer import ContactNetsTrainer
    g = np.load(os.path.join(GOLDEN_DIR, 'cube_box_4096.npz'))
    x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')

    def run(use_graph):
        system = MultibodyLearnableSystem({'cube': os.path.join(ASSET_DIR, 'cube.urdf')}, float(g['dt']),
                                          dtype=torch.float32, device='cuda:0')
        with torch.no_grad():
            system.multibody_terms.contact_terms.geometries[1].length_params.mul_(1.2)
        trainer = ContactNetsTrainer(system, lr=1e-3, batch_size=1000, use_graph=use_graph)  # 4 full batches + a tail of 96
        trainer.train_epoch(x, xp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        log = trainer.fit(x, xp, epochs=4)
        torch.cuda.synchronize()
        # how the 5 epochs x (4 full batches + a tail) reached the device: eagerly every step is enqueued kernel by kernel; with
        # the graph only the warm-up, the capture and the tails are -- every full batch is ONE replay
        expect = {'enqueued': 5 * 5, 'replayed': 0} if not use_graph else {'enqueued': 2 + 5, 'replayed': 5 * 4}
        assert trainer.step_counts == expect, (use_graph, trainer.step_counts)
        return system, log, (time.perf_counter() - t0) / (4 * 5)

    eager, log_e, t_e = run(False)
    graph, log_g, t_g = run(True)
    for (name, a), (_, b) in zip(eager.named_parameters(), graph.named_parameters()):
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-7), name
    assert np.allclose(log_e.epoch_losses, log_g.epoch_losses, rtol=1e-4)
    print(f'training step: eager {t_e * 1e6:.0f} us, graph {t_g * 1e6:.0f} us (reported, not asserted: wall time of 20 steps is noise)')


def test_gradient_buffers_do_not_alias_across_paths():
    """(1) accumulate=True must add the new gradient to the old one although the old `.grad` is a view of
    the buffer the next launch overwrites; (2) an autograd backward after a fused step (no zero_grad, or
    zero_grad(set_to_none=False)) must add its gradient once -- the kernel must not write into `.grad` itself."""
    from dair_pll_amd import MultibodyLearnableSystem
    g = np.load(os.path.join(GOLDEN_DIR, 'cube_box_4096.npz'))
    system = MultibodyLearnableSystem({'cube': os.path.join(ASSET_DIR, 'cube.urdf')}, float(g['dt']), dtype=torch.float64,
                                      device='cuda:0')
    xa, xpa = (torch.tensor(g[k][:512], device='cuda:0') for k in ('x', 'x_plus'))
    xb, xpb = (torch.tensor(g[k][512:1024], device='cuda:0') for k in ('x', 'x_plus'))
    flat = lambda: torch.cat([p.grad.reshape(-1) for p in system._param_list()]).clone()
    system.contactnets_loss_and_grad(xa, xpa)
    ga = flat()
    system.zero_grad()
    system.contactnets_loss_and_grad(xb, xpb)
    gb = flat()
    # (1) accumulate
    system.zero_grad()
    system.contactnets_loss_and_grad(xa, xpa)
    system.contactnets_loss_and_grad(xb, xpb, accumulate=True)
    assert (flat() - (ga + gb)).abs().max() <= 1e-15 * (ga + gb).abs().max()
    # (2) fused step, then autograd on top without clearing
    system.zero_grad()
    system.contactnets_loss_and_grad(xa, xpa)
    system.contactnets_loss(xb, torch.zeros((512, 0), device='cuda:0'), xpb).mean().backward()
    assert (flat() - (ga + gb)).abs().max() <= 1e-14 * (ga + gb).abs().max()
    # ... and with gradients zeroed in place in between
    system.zero_grad(set_to_none=False)
    system.contactnets_loss(xb, torch.zeros((512, 0), device='cuda:0'), xpb).mean().backward()
    assert (flat() - gb).abs().max() <= 1e-14 * gb.abs().max()


def test_empty_shard_contributes_a_zero_row():
    """A ragged tail batch smaller than the world size leaves some ranks an empty shard: with `global_batch` set the
    call launches no item workgroups, writes zero loss / gradients and (distributed) still takes part in the exchange."""
    from dair_pll_amd import MultibodyLearnableSystem, _capi
    g = np.load(os.path.join(GOLDEN_DIR, 'cube_box_4096.npz'))
    system = MultibodyLearnableSystem({'cube': os.path.join(ASSET_DIR, 'cube.urdf')}, float(g['dt']), dtype=torch.float32,
                                      device='cuda:0')
    x = torch.tensor(g['x'][:8], dtype=torch.float32, device='cuda:0')
    xp = torch.tensor(g['x_plus'][:8], dtype=torch.float32, device='cuda:0')
    system.contactnets_loss_and_grad(x, xp)
    assert system.grad_buffer().abs().max() > 0
    system.global_batch = 3
    total = system.contactnets_loss_and_grad(x[:0], xp[:0])
    assert total.item() == 0.0 and torch.count_nonzero(system.grad_buffer()).item() == 0
    system.global_batch = 0
    with pytest.raises(_capi.DpllError):
        system.contactnets_loss_and_grad(x[:0], xp[:0])


@pytest.mark.parametrize('use_graph', [False, True])
def test_training_a_general_model_with_a_body_body_pair(use_graph):
    """The general build under the trainer (clasp_ball: a sphere base, a polygon tip, one body-body candidate): the data are
    the model's own rollouts, the student starts with a wrong radius, jittered vertices and wrong frictions; the fused
    loss + Adam (parameters are views of the packed [theta | friction (5) | geometry blocks (4, 24)] buffer, eager and
    replayed as a hipGraph) reduce the loss and move the radius back."""
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.trainer import ContactNetsTrainer
    g = np.load(os.path.join(GOLDEN_DIR, 'clasp_ball_literal.npz'))
    x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
    system = MultibodyLearnableSystem({'clasp_ball': os.path.join(ASSET_DIR, 'clasp_ball.urdf')}, float(g['dt']), dtype=torch.float32,
                                      device='cuda:0', mesh_representation='polygon')
    geometries = system.multibody_terms.contact_terms.geometries
    true_radius = geometries[1].length_param.item()
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        geometries[1].length_param.mul_(1.3)
        noise = torch.randn(geometries[2].vertices.shape, generator=gen, dtype=torch.float32)  # (dtype: other test modules set the default)
        geometries[2].vertices.add_(0.001 * noise.to('cuda:0'))
        system.multibody_terms.contact_terms.friction_params.mul_(1.5)
    start_radius = geometries[1].length_param.item()
    # Adam moves every parameter by about lr per step whatever its scale: 2e-4 m per step for lengths of centimetres
    trainer = ContactNetsTrainer(system, lr=2e-4, batch_size=x.shape[0], use_graph=use_graph)
    log = trainer.fit(x, xp, epochs=60)
    assert np.isfinite(log.epoch_losses).all() and log.epoch_losses[-1] < 0.8 * log.epoch_losses[0]
    assert abs(geometries[1].length_param.item() - true_radius) < abs(start_radius - true_radius)
    # the parameters are still views of the packed buffer the kernels read
    flat = system._packed()
    assert geometries[2].vertices.data_ptr() == flat.data_ptr() + (30 + 5 + 24) * flat.element_size()


@pytest.mark.parametrize('use_graph', [False, True])
def test_training_a_model_with_a_prismatic_joint_and_turned_frames(use_graph):
    """slider.urdf under the trainer (a prismatic joint, rpy on the joint and on a collision box): the data are the
    model's own rollouts (the reference-run fixture), the student starts with the carriage box 25 % too large, the knob's
    radius 20 % too small and wrong masses; the fused loss + Adam, eager and replayed as a hipGraph, reduce the loss and
    move both shapes back towards the truth."""
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.trainer import ContactNetsTrainer
    g = np.load(os.path.join(GOLDEN_DIR, 'slider_literal.npz'))
    x = torch.tensor(g['x'], dtype=torch.float32, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=torch.float32, device='cuda:0')
    system = MultibodyLearnableSystem({'slider': os.path.join(ASSET_DIR, 'slider.urdf')}, float(g['dt']), dtype=torch.float32, device='cuda:0')
    assert system.spec.bodies[1].joint_kind == 'prismatic' and system.spec.rotated()
    geometries = system.multibody_terms.contact_terms.geometries
    true_box, true_radius = geometries[2].length_params.detach().clone(), geometries[3].length_param.item()
    with torch.no_grad():
        geometries[2].length_params.mul_(1.25)
        geometries[3].length_param.mul_(0.8)
        system.multibody_terms.lagrangian_terms.inertial_parameters[:, 0].add_(0.1)
    start_box, start_radius = geometries[2].length_params.detach().clone(), geometries[3].length_param.item()
    trainer = ContactNetsTrainer(system, lr=2e-4, batch_size=x.shape[0], use_graph=use_graph)
    log = trainer.fit(x, xp, epochs=60)
    assert np.isfinite(log.epoch_losses).all() and log.epoch_losses[-1] < 0.8 * log.epoch_losses[0]
    assert (geometries[2].length_params.detach().abs() - true_box.abs()).abs().sum() < (start_box.abs() - true_box.abs()).abs().sum()
    assert abs(abs(geometries[3].length_param.item()) - true_radius) < abs(start_radius - true_radius)


@pytest.mark.parametrize('model', ['cube', 'elbow'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_fused_adam_step_is_torch_adam(model, dtype):
    """dpll_contactnets_train_step: the finalize kernel of the loss launch applies Adam in place.  Forty steps with
    weight decay against torch.optim.Adam (experiment.py:213-228) fed by contactnets_loss_and_grad, eagerly and as a
    replayed hipGraph: same parameters, same moments, same step count."""
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.trainer import ContactNetsTrainer
    g = np.load(os.path.join(GOLDEN_DIR, 'cube_box_4096.npz' if model == 'cube' else 'elbow_box_4096.npz'))
    x = torch.tensor(g['x'][:2048], dtype=dtype, device='cuda:0')
    xp = torch.tensor(g['x_plus'][:2048], dtype=dtype, device='cuda:0')

    def make():
        system = MultibodyLearnableSystem({model: os.path.join(ASSET_DIR, model + '.urdf')}, float(g['dt']), dtype=dtype, device='cuda:0')
        with torch.no_grad():
            system.multibody_terms.contact_terms.geometries[1].length_params.mul_(1.2)
            system.multibody_terms.contact_terms.friction_params[1] = 0.5
        return system

    ref, eager, graphed = make(), make(), make()
    optimizer = torch.optim.Adam(ref.parameters(), lr=2e-3, weight_decay=1e-3)
    t_eager = ContactNetsTrainer(eager, lr=2e-3, weight_decay=1e-3, batch_size=512, seed=3, fused_adam=True)
    t_graph = ContactNetsTrainer(graphed, lr=2e-3, weight_decay=1e-3, batch_size=512, seed=3, fused_adam=True, use_graph=True)
    order_gen = torch.Generator().manual_seed(3)
    for epoch in range(10):
        order = torch.randperm(x.shape[0], generator=order_gen).to(x.device)  # the trainer's own shuffle, replayed for the reference
        for start in range(0, x.shape[0], 512):
            idx = order[start:start + 512]
            ref.contactnets_loss_and_grad(x[idx], xp[idx])
            optimizer.step()
        loss_eager = t_eager.train_epoch(x, xp)
        loss_graph = t_graph.train_epoch(x, xp)
        assert abs(loss_eager - loss_graph) <= 1e-6 * max(1.0, abs(loss_eager))
    tol = 2e-5 if dtype == torch.float32 else 1e-10
    for (name, p_ref), (_, p_eager), (_, p_graph) in zip(ref.named_parameters(), eager.named_parameters(), graphed.named_parameters()):
        scale = max(1.0, p_ref.abs().max().item())
        assert (p_ref - p_eager).abs().max().item() <= tol * scale, (name, (p_ref - p_eager).abs().max().item())
        assert (p_eager - p_graph).abs().max().item() <= tol * scale, name
        assert (p_ref - make().get_parameter(name)).abs().max().item() > 1e-4  # (the parameters did move)
    assert t_eager.optimizer.step.item() == 40.0 and t_graph.optimizer.step.item() == 40.0
    state = optimizer.state[next(iter(ref.parameters()))]
    n = state['exp_avg'].numel()
    assert torch.allclose(t_eager.optimizer.exp_avg[:n].reshape(state['exp_avg'].shape), state['exp_avg'], rtol=1e-4, atol=1e-9)


FUSED_CASES = {
    # name -> (init_urdfs, fixture, mesh_representation): the general build (a prismatic joint + turned frames; a body-body pair), the
    # forest build (two models in one system), the specialised mesh build (every weight of the network is a parameter)
    'slider': ({'slider': 'slider.urdf'}, 'slider_literal', 'deep_support'),
    'clasp': ({'clasp': 'clasp.urdf'}, 'clasp_literal', 'deep_support'),
    'two_cubes': ({'cube_a': 'cube.urdf', 'cube_b': 'cube.urdf'}, 'two_cubes_literal', 'deep_support'),
    'cube_mesh': ({'cube': 'cube_mesh.urdf'}, 'cube_mesh_literal', 'deep_support'),
    # a general tree with two learned shapes and a candidate between them: head by the general finalize kernel, each network's
    # weights by its own reduce kernel, the optimizer state moved on by the last launch
    'clasp_mesh': ({'clasp_mesh': 'clasp_mesh.urdf'}, 'clasp_mesh_literal', 'deep_support'),
}


@pytest.mark.parametrize('case', list(FUSED_CASES))
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_fused_adam_step_of_the_other_builds_is_torch_adam(case, dtype):
    """dpll_contactnets_train_step on the general and the forest build (Adam in the kernel that chains the folded rows; padding
    entries of the flat buffer untouched) and dpll_contactnets_train_step_mesh on the specialised mesh build (Adam where the
    network's weight gradients are reduced): twelve steps with weight decay against torch.optim.Adam fed by
    contactnets_loss_and_grad -- same parameters, same step count; eagerly and as a replayed hipGraph."""
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.system import FusedAdamState
    urdfs, fixture, representation = FUSED_CASES[case]
    g = np.load(os.path.join(GOLDEN_DIR, fixture + '.npz'))
    x = torch.tensor(g['x'], dtype=dtype, device='cuda:0')
    xp = torch.tensor(g['x_plus'], dtype=dtype, device='cuda:0')

    def make():
        torch.manual_seed(0)  # (the mesh system draws its network at construction)
        system = MultibodyLearnableSystem({k: os.path.join(ASSET_DIR, v) for k, v in urdfs.items()}, float(g['dt']), dtype=dtype, device='cuda:0',
                                          mesh_representation=representation)
        with torch.no_grad():
            system.multibody_terms.contact_terms.friction_params[1] = 0.5
        return system

    ref, fused, graphed = make(), make(), make()
    flat0 = fused._packed().clone()
    # (learned shapes: a step far shorter than the networks' weights -- about 2e-3 -- so that few of them can cross zero, see below)
    lr = 2e-5 if 'mesh' in case else 1e-3
    optimizer = torch.optim.Adam(ref.parameters(), lr=lr, weight_decay=1e-3)
    adam, adam_g = FusedAdamState(lr=lr, weight_decay=1e-3), FusedAdamState(lr=lr, weight_decay=1e-3)
    graphed.contactnets_train_step(x, xp, adam_g)  # warm-up of the path to be captured, undone below
    with torch.no_grad():
        graphed._packed().copy_(flat0)
        for t, value in zip((adam_g.exp_avg, adam_g.exp_avg_sq), (0.0, 0.0)):
            t.fill_(value)
        adam_g.state.copy_(torch.tensor([0.0, 1.0, 1.0], dtype=torch.float64))
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            graphed.contactnets_train_step(x, xp, adam_g)
    torch.cuda.current_stream().wait_stream(side)
    with torch.no_grad():  # (capture does not execute: nothing to undo)
        assert torch.equal(graphed._packed(), flat0)
    for step in range(12):
        ref.contactnets_loss_and_grad(x, xp)
        if step == 0:
            first_grad = torch.cat([p.grad.reshape(-1) for p in ref.parameters()]).clone()
        optimizer.step()
        fused.contactnets_train_step(x, xp, adam)
        if step == 0:  # same parameters, same kernels: the gradient the fused update used IS the one loss_and_grad returns
            fused_grad = torch.cat([p.grad.reshape(-1) for p in fused.parameters()])
            worst = (first_grad - fused_grad).abs().argmax().item()
            assert torch.equal(first_grad, fused_grad), (worst, first_grad[worst].item(), fused_grad[worst].item())
        graph.replay()
    torch.cuda.synchronize()
    tol = 3e-5 if dtype == torch.float32 else 1e-9
    moved = 0.0
    for (name, p_ref), (_, p_fused), (_, p_graph), (_, p_start) in zip(ref.named_parameters(), fused.named_parameters(), graphed.named_parameters(),
                                                                       make().named_parameters()):
        scale = max(1.0, p_ref.abs().max().item())
        diff = (p_ref - p_fused).abs()
        if 'hidden_weights' in name or 'output_weight' in name:
            # these weights enter the network through their ABSOLUTE value: a weight that the twelve steps (each at most a few
            # lr long) can carry across zero sees the sign of its gradient flip there, and on which side of zero a step lands is
            # decided by the last bit -- torch's float32 Adam and the kernel's (double inside) part ways by up to the steps
            # that follow.  Held to the tolerance: every weight that cannot reach zero; the others to the length of the path.
            reach = 12 * 3 * lr
            far = p_start.abs() > reach
            assert far.float().mean().item() > 0.5, name  # (the check must not be vacuous)
            assert diff[far].max().item() <= tol * scale, (name, diff[far].max().item())
            assert diff.max().item() <= 2 * reach, (name, diff.max().item())
        else:
            assert diff.max().item() <= tol * scale, (name, diff.max().item())
        # (eager and replayed: the same launches on the same numbers)
        assert (p_fused - p_graph).abs().max().item() <= tol * scale, name
        moved = max(moved, (p_ref - p_start).abs().max().item())
    assert moved > 5 * lr and adam.step.item() == 12.0 and adam_g.step.item() == 12.0
    # what is not a parameter stays what it was: padding of the flat buffer (general build: unused geometry slots; every build
    # but the mesh one: the tail of each geometry's block)
    layout, total = fused._layout()
    real = torch.zeros(total, dtype=torch.bool)
    for p, offset in layout:
        real[offset:offset + p.numel()] = True
    assert torch.equal(fused._packed().cpu()[~real], flat0.cpu()[~real])


def test_fused_training_step_refuses_what_it_cannot_do():
    """a multi-process gradient exchange inside the fused step of a mesh system is not built: refused, not skipped silently"""
    from dair_pll_amd import MultibodyLearnableSystem
    from dair_pll_amd.system import FusedAdamState
    system = MultibodyLearnableSystem({'m': os.path.join(ASSET_DIR, 'clasp_mesh.urdf')}, 0.0068, device='cuda:0')
    x = torch.zeros((4, system.space.n_x), device='cuda:0')
    system.grad_world = 2
    with pytest.raises(NotImplementedError):
        system.contactnets_train_step(x, x, FusedAdamState())
