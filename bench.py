#!/usr/bin/env python
"""
Headline benchmark: env-steps/sec of the PPO learner ingest on BASELINE.json's workload
(configs[4] = "PPO synthetic 1024 actors x 128 steps x 376-dim obs"), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one PPOLearner.learn() over one batch of B = 1024 sub-trajectories x N = 128 steps
per GPU that is ALREADY RESIDENT IN HBM: critic pass over all B*(N+1) steps, windowed GAE,
advantage normalisation, 10 policy epochs + 10 value epochs (forward, loss, backward,
clip_grad_norm_, Adam), z-filter update, statistics read-back.  The KL early exit is disabled
(kl_target = 1e9) so that every step does the full 10 + 10 epochs -- the same setting is applied
to the CPU baseline.  Weak scaling: each rank owns its own 1024
sub-trajectories (actors shard across GPUs); gradients / loss sums / moments are all-reduced.

Prints ONE JSON line on rank 0 (see the driver contract in the task description), with
`roofline` for the dominant kernel (the fused FP32-MFMA critic pass) and `cpu_baseline`
(oracle/ppo_oracle.py -- the CPU restatement of the reference learner, pinned bit-for-bit
against the reference's own code -- timed on this host's cores).
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from surreal_amd import synthetic  # noqa: E402
from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config  # noqa: E402

B, N, D, A = 1024, 128, 376, 17
HIDDEN = (300, 200)
METRIC = 'env-steps/sec (learner ingest) PPO 1024 actors×128 steps, 1/2/4/8 MI355X'
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md (dense FP32 MFMA)
PEAK_HBM_GBPS = 8000.0


def algorithmic_costs(split_tail):
    """per-launch figures of the fused critic kernel.  With the tail
    split the launch covers the B*N step rows (the B obs_next rows go through the layered
    kernels): 1 508 B and 346 kFLOP per row either way."""
    rows = B * N if split_tail else B * (N + 1)
    flops = 2.0 * rows * (D * HIDDEN[0] + HIDDEN[0] * HIDDEN[1] + HIDDEN[1] * 1)
    bytes_ = 4.0 * (rows * D + rows)          # read obs||obs_next once, write values
    return rows, flops, bytes_


def build_learner(mode, device_index):
    from surreal_amd.learner.ppo import PPOLearner
    lc = ppo_learner_config()
    lc.model.actor_fc_hidden_sizes = list(HIDDEN)
    lc.model.critic_fc_hidden_sizes = list(HIDDEN)
    lc.algo.n_step = N
    lc.algo.stride = N
    lc.algo.rnn.if_rnn_policy = False
    lc.algo.ppo_mode = mode
    lc.algo.consts.kl_target = 1e9          # no data-dependent early exit: full 10 + 10 epochs
    lc.replay.batch_size = B
    learner = PPOLearner(lc, ppo_env_config(D, A), ppo_session_config('/tmp/surreal_amd_bench'))
    params = synthetic.make_ppo_params(D, A, hidden=HIDDEN, seed=1)
    zstate = synthetic.make_zfilter_state(D, seed=2)
    learner.model.load_params(params)
    learner.ref_target_model.load_params(params)
    learner.model.z_filter.load_state_dict(zstate)
    learner.ref_target_model.z_filter.load_state_dict(zstate)
    return learner, params, zstate


def device_batch(learner, rank):
    """synthetic batch of the BASELINE shape, generated once and left resident in HBM"""
    batch = synthetic.make_ppo_batch(B, N, D, A, seed=100 + rank)
    return learner._preprocess_batch_ppo(copy.deepcopy(batch)), batch


def time_fused_kernel(learner, dbatch, iters=10):
    """average duration of the dominant kernel: HIP events on the launch stream around ONE launch
    issued right after a full learn() step (same clocks / cache state as inside the timed steps;
    back-to-back launches of this MFMA-saturating kernel run ~7 % slower under the power cap)"""
    ws = learner._ws
    obs = dbatch['obs']['low_dim']['flat_inputs']
    obs_next = dbatch['obs_next']['low_dim']['flat_inputs']
    m = learner.model
    K = learner.K
    tail = None if ws.split_tail else obs_next       # exactly the launch learn() issues
    out = ws.vals[:B * N] if ws.split_tail else ws.vals
    total = 0.0
    for _ in range(iters):
        learner.learn(dbatch)
        zm, zs = m.z_filter.refresh_stats()
        K.mlp3_pack(m.critic, ws.packed)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.mlp3_forward_fused(ws.packed, m.critic, obs, tail, zm, zs, out, 0)
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1)
    return total / iters * 1e-3


def measured_traffic(rows):
    """HBM bytes per launch of the fused kernel from the committed PMC pass (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 correction applied; profiles/*_pmc_*.json).
    PMC collection cannot run inside this process, so the figure is read from the profile."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_hbm_traffic.json')), reverse=True):
        try:
            fk = json.load(open(path)).get('fused_kernel', {})
        except Exception:
            continue
        for v in fk.values():
            if v.get('rows_upper_bound') == rows:
                return v['hbm_read_bytes_corrected'] + v['hbm_write_bytes'], os.path.basename(path)
    return None, None


def measured_mfma_util(rows):
    """MFMA-busy share of the fused kernel from the committed PMC pass (profiles/*_pmc_mfma.json)"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_mfma.json')), reverse=True):
        try:
            ks = json.load(open(path)).get('kernels', {})
        except Exception:
            continue
        for k, v in ks.items():
            if 'fused' in k and k.endswith('@grid%d' % (rows // 128 * 256)) and 'mfma_util_pct' in v:
                return v['mfma_util_pct'], os.path.basename(path)
    return None, None


def cpu_baseline(mode, params, zstate, batch, budget_s=20.0):
    """the reference learner's CPU path (oracle restatement, same ATen ops) on this host"""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ppo_oracle
    cores = torch.get_num_threads()
    O = ppo_oracle.OraclePPOLearner(params, A, B, zstate=zstate, n_step=N, ppo_mode=mode,
                                    kl_target=1e9)
    O.learn(copy.deepcopy(batch))        # warm-up
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < budget_s and n < 50):
        O.learn(copy.deepcopy(batch))
        n += 1
        if time.time() - t0 > budget_s:
            break
    dt = (time.time() - t0) / n
    return {'value': B * N / dt, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'sample': '%d learn() calls of the full 1024x128x376 batch (10+10 epochs, %s mode), '
                      '%.3f s each, torch %s CPU' % (n, mode, dt, torch.__version__)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--mode', default='adapt', choices=['adapt', 'clip'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--split-chains', action='store_true', help='policy and value epochs on two streams')
    ap.add_argument('--schedule', default='lockstep', choices=['lockstep', 'two_stream'],
                    help='epoch launch schedule (session_config.learner.epoch_schedule)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    # SMX_BENCH_BACKEND=gloo is the single-GPU rehearsal of the N > 1 path (tests/test_gpu_dist.py):
    # RCCL refuses two ranks on one device, gloo does not, and the ranks then share the GPU
    backend = os.environ.get('SMX_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    learner, params, zstate = build_learner(args.mode, local_rank)
    learner.epoch_schedule = args.schedule
    if args.no_graph:
        learner.use_graph = False
    dbatch, batch = device_batch(learner, rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        learner.learn(dbatch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = learner.learn(dbatch)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert learner.epochs_executed == learner.epoch_policy, 'work was skipped inside the timed region'

    kt = time_fused_kernel(learner, dbatch)
    out = None
    if rank == 0:
        rows, flops, bytes_ = algorithmic_costs(learner._ws.split_tail)
        out = {
            'metric': METRIC,
            'value': world * B * N * args.steps / dt,
            'unit': 'env-steps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': 'BASELINE configs[4]: PPO synthetic 1024 actors x 128 steps x 376-dim '
                            'obs per GPU, A=17, MLP [300,200], z-filter, %s mode, 10 policy + 10 '
                            'value epochs (KL early exit disabled), batch resident in HBM' % args.mode,
                'B_per_gpu': B, 'n_step': N, 'obs_dim': D, 'action_dim': A,
                'hip_graph': bool(learner.use_graph), 'parallelism': 'dp%d' % world,
            },
            'roofline': {
                'kernel': 'mlp3_fused_kernel<10,7,true> (z-filter + critic MLP over %d rows)' % rows,
                'bound': 'mfma',
                'achieved': flops / kt / 1e12,
                'peak': PEAK_FP32_MFMA_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': flops / kt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                'traffic': measured_traffic(rows)[0],
                'traffic_source': measured_traffic(rows)[1],
                'mfma_busy_pct': measured_mfma_util(rows)[0],
                'mfma_busy_source': measured_mfma_util(rows)[1],
                'kernel_ms': kt * 1e3,
                'flops_per_launch': flops,
                'algorithmic_bytes_per_launch': bytes_,
                'hbm_GBps': bytes_ / kt / 1e9,
                'hbm_frac': bytes_ / kt / 1e9 / PEAK_HBM_GBPS,
                'share_of_step': kt / (dt / args.steps),
            },
            'final_stats': {k: stats[k] for k in ('_surr_loss', '_val_loss', '_pol_kl') if k in stats},
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.mode, params, zstate, batch)
            out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
