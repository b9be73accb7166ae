#!/usr/bin/env python
"""
Secondary measurement (SURVEY.md 8(d) "secondary", rows a13/a16-a18 of 8(a)): the WHOLE on-device
loop of one GPU -- n actors stepping the synthetic environment under the current policy
(PPOAgent.act_batch + SyntheticVecEnv.step, T steps), the moving-window cut
(exp_sender_wrapper.py:209-228), FIFO insert + pop (fifo_replay.py:6-49, device tier) and
PPOLearner.learn -- with per-stage timings.  Prints one JSON line.

    python scripts/bench_pipeline.py [--actors 1024] [--steps 128] [--iters 10] [--graph] [--fused-step]

--actors may exceed --learn-batch (1024 sub-trajectories per learn, the benchmark configuration): one rollout then
feeds actors / learn-batch learner iterations through the FIFO.  Acting is a chain of dependent launches per
environment step whose cost barely depends on the number of actors, so more actors per GPU amortise it -- the
reference's deployment knob too (agents per learner).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_pipeline(actors=1024, steps=128, obs_dim=376, action_dim=17, iters=10, warmup=3, graph=True, fused_step=True,
                 copies=False, cpu_double=False, learn_batch=1024, pixel=None, frame_stacks=1, rnn=False, overlap=False,
                 no_zero_copy=False):
    """-> the result dict main() prints (also called by bench.py for its `secondary` entry)"""
    args = argparse.Namespace(actors=actors, steps=steps, obs_dim=obs_dim, action_dim=action_dim, iters=iters,
                              warmup=warmup, graph=graph, fused_step=fused_step, copies=copies,
                              cpu_double=cpu_double, learn_batch=learn_batch, pixel=pixel, frame_stacks=frame_stacks,
                              rnn=rnn, overlap=overlap, no_zero_copy=no_zero_copy)
    return _run(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--actors', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=128)
    ap.add_argument('--obs-dim', type=int, default=376)
    ap.add_argument('--action-dim', type=int, default=17)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--graph', action='store_true', help='replay the rollout as one hipGraph')
    ap.add_argument('--fused-step', action='store_true',
                    help='sample + env step + next z-filter in one launch (SyntheticVecEnv.rollout)')
    ap.add_argument('--copies', action='store_true', help='insert / pop through copies (no table views)')
    ap.add_argument('--cpu-double', action='store_true', help='dry run on the CPU test double')
    ap.add_argument('--learn-batch', type=int, default=1024, help='sub-trajectories per learner iteration')
    ap.add_argument('--pixel', type=int, nargs=3, default=None, metavar=('C', 'H', 'W'),
                    help='camera frames (uint8) next to the low-dim state: the CNN-stem policy (configs[3]: 3 84 84)')
    ap.add_argument('--frame-stacks', type=int, default=1, help='frames the policy sees, stacked on the channel axis')
    ap.add_argument('--rnn', action='store_true', help='LSTM-stem policy (the reference default)')
    ap.add_argument('--no-zero-copy', action='store_true', help='rollout -> window cut -> FIFO as separate launches')
    ap.add_argument('--overlap', action='store_true',
                    help='actors one rollout ahead of the learner: rollout k + 1 on a second stream while learn k runs '
                         '(how the reference runs: agents keep acting on the parameters they last fetched)')
    args = ap.parse_args()
    print(json.dumps(_run(args)), flush=True)


def _run(args):
    from surreal_amd import kernels as KN
    if args.cpu_double:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from cpu_kernels import TorchCpuKernels
        KN.set_default_kernels(TorchCpuKernels(), 'cpu')
    from surreal_amd.agent import PPOAgent
    from surreal_amd.env import SyntheticVecEnv
    from surreal_amd.learner import PPOLearner
    from surreal_amd.replay import FIFOReplay
    from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config
    n, T, D, A = args.actors, args.steps, args.obs_dim, args.action_dim
    LB = min(args.learn_batch, n)
    assert n % LB == 0, 'actors must be a multiple of the learner batch'
    lc = ppo_learner_config()
    lc.algo.n_step, lc.algo.stride = T, T
    lc.algo.rnn.if_rnn_policy = bool(args.rnn)
    stem = bool(args.rnn) or args.pixel is not None
    lc.algo.consts.kl_target = 1e9                      # no early exit: every learn does 10 + 10 epochs
    lc.replay.batch_size, lc.replay.memory_size = LB, 2 * n
    cam = (args.frame_stacks * args.pixel[0], args.pixel[1], args.pixel[2]) if args.pixel else None   # what the policy sees
    ec, sc = ppo_env_config(D, A, pixel=cam), ppo_session_config()
    # actors one rollout ahead: the rollout kernel of the side stream holds CUs while learn() runs -- the learner must
    # not use the epoch launch that waits inside itself for all of its workgroups (learner/ppo.py: exclusive_device)
    sc.learner.exclusive_device = not bool(args.overlap)
    learner = PPOLearner(lc, ec, sc)
    learner.graph_input_sets = 2 * max(1, n // LB)      # the FIFO hands out batches from two alternating row ranges
    agent = PPOAgent(lc, ec, sc, agent_id=0, agent_mode='training')
    agent.attach_learner(learner)
    agent.fetch_parameter()
    replay = FIFOReplay(lc, ec, sc)
    venv = SyntheticVecEnv(n, D, A, episode_len=T, pixel=tuple(args.pixel) if args.pixel else None,
                           frame_stacks=args.frame_stacks)
    dev = venv.device
    sync = torch.cuda.synchronize if str(dev).startswith('cuda') else (lambda: None)

    venv.start_rollout(T, info_width=2 * A)

    act_buf = torch.empty(n, A, device=dev)

    def rollout_body():
        if args.fused_step:                                # 4 launches per env step
            venv.rollout(agent)
            return
        eps = torch.randn(T, n, A, device=dev)             # the whole rollout's noise in one launch
        for t in range(T):
            pd_slot = venv.rolls['pds'][:, t]
            agent.act_batch(venv.state, eps=eps[t], out_actions=act_buf, out_pd=pd_slot)
            venv.step(act_buf, pds=pd_slot)

    graph = None

    def rollout():
        venv.reset()
        venv.slot = 0
        if graph is not None:
            graph.replay()
            venv.slot, venv.t = T, 0
        else:
            rollout_body()

    to_batch = venv.to_batch
    use_graph = bool(args.graph) and not stem        # (the stem path allocates per step: eager launches)
    # stride == n_step: a window is the rollout -- the one-launch kernel records straight into the FIFO's slots
    # (SyntheticVecEnv.rollout_into): no window cut, no insert copy, the learner pops views
    zero_copy = bool(args.fused_step) and not args.copies and not stem and not args.no_zero_copy and venv.can_rollout_into(agent)

    def rollout_to_fifo():
        slots = replay.reserve_batch(n, venv.window_shapes(T))
        if slots is None:
            return False
        venv.reset()
        venv.rollout_into(agent, slots)
        replay.commit_batch(n)
        return True

    def iteration(times=None):
        t0 = time.perf_counter()
        direct = zero_copy and rollout_to_fifo()
        if not direct:
            rollout()
        if times is not None:
            sync()
        t1 = time.perf_counter()
        slots = None if (args.copies or stem or direct) else replay.reserve_batch(n, venv.window_shapes(T))
        if direct:
            pass
        elif slots is not None:                  # windows are cut straight into the FIFO table
            venv.emit_windows(T, T, out=slots)
            replay.commit_batch(n)
        else:
            replay.insert_batch(venv.emit_windows(T, T))
        if times is not None:
            sync()
        t2 = time.perf_counter()
        for _ in range(n // LB):               # the FIFO hands the rollout over one learner batch at a time
            if times is not None:
                sync()
            tb = time.perf_counter()
            batch = replay.sample_batch(LB, copy=bool(args.copies))
            if times is not None:
                sync()
            t2 += time.perf_counter() - tb          # (t2 - t1 = window cut + FIFO hand-over)
            learner.learn(to_batch(batch))
        agent.fetch_parameter()
        if times is not None:
            sync()
            t3 = time.perf_counter()
            times.append((t1 - t0, t2 - t1, t3 - t0 - (t1 - t0) - (t2 - t1)))

    iteration()
    if use_graph and not args.cpu_double and not zero_copy:
        import gc
        sync()
        venv.reset()
        venv.slot = 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            rollout_body()                                # warm every lazily created buffer
        torch.cuda.current_stream().wait_stream(side)
        sync()
        venv.reset()
        venv.slot = 0
        gc.collect()
        gc.disable()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                rollout_body()
            graph = g
        finally:
            gc.enable()
    if args.overlap and not args.cpu_double:
        # actors one rollout ahead of the learner: while learn(k) runs on the main stream, rollout k + 1 runs on a
        # second one under the parameters the agent fetched last (surreal's agents act asynchronously in the same way,
        # agent/base.py:182-198); the window cut of rollout k sits between the two (it reads the rollout tables)
        side = torch.cuda.Stream()
        sequential = iteration

        def iteration(times=None):                  # noqa: F811
            main_s = torch.cuda.current_stream()
            if zero_copy:
                # rollout k sits in the FIFO already; rollout k + 1 goes into the NEXT slots on the side stream while
                # the learner pops and learns rollout k
                side.wait_stream(main_s)
                batches = [replay.sample_batch(LB, copy=False) for _ in range(n // LB)]
                with torch.cuda.stream(side):
                    ok = rollout_to_fifo()
                    assert ok, 'the FIFO must hold two rollouts (memory_size >= 2 * actors)'
                for b in batches:
                    learner.learn(to_batch(b))
                main_s.wait_stream(side)
                agent.fetch_parameter()
                if times is not None:
                    sync()
                    times.append((0.0, 0.0, 0.0))
                return
            slots = None if (args.copies or stem) else replay.reserve_batch(n, venv.window_shapes(T))
            if slots is not None:
                venv.emit_windows(T, T, out=slots)
                replay.commit_batch(n)
            else:
                replay.insert_batch(venv.emit_windows(T, T))
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                rollout()
            for _ in range(n // LB):
                learner.learn(to_batch(replay.sample_batch(LB, copy=bool(args.copies))))
            main_s.wait_stream(side)
            agent.fetch_parameter()
            if times is not None:
                sync()
                times.append((0.0, 0.0, 0.0))
        if zero_copy:
            assert rollout_to_fifo()                # rollout 0: the pipeline's fill
        else:
            rollout()                               # rollout 0: the pipeline's fill
    for _ in range(args.warmup):
        iteration()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        iteration()
    sync()
    whole = (time.perf_counter() - t0) / args.iters
    staged = []
    for _ in range(args.iters):
        iteration(staged)
    st = [sum(x[i] for x in staged) / len(staged) for i in range(3)]
    out = {'metric': 'env-steps/s, whole on-device loop (act + env step + windows + FIFO + learn)',
           'value': n * T / whole, 'ms_per_iteration': whole * 1e3,
           'config': {'actors': n, 'steps_per_rollout': T, 'obs_dim': D, 'action_dim': A, 'learn_batch': LB,
                      'learns_per_rollout': n // LB, 'rollout_graph': graph is not None,
                      'fused_step': bool(args.fused_step), 'overlap': bool(args.overlap), 'rnn': bool(args.rnn),
                      'rollout_into_fifo_slots': bool(zero_copy),
                      'pixel': list(args.pixel) if args.pixel else None, 'frame_stacks': args.frame_stacks},
           'stage_ms_synchronised': {'rollout': st[0] * 1e3, 'windows+fifo': st[1] * 1e3, 'learn': st[2] * 1e3},
           'rollout_env_steps_per_s': n * T / st[0] if st[0] > 0 else None}
    return out


if __name__ == '__main__':
    main()
