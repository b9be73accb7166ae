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
`roofline` for the dominant kernel (the fused FP32-MFMA critic pass), `step_roofline` for the whole
learn() (71 GFLOP and 200.5 MB algorithmic per learn), `cpu_baseline`
(oracle/ppo_oracle.py -- the CPU restatement of the reference learner, pinned bit-for-bit against
the reference's own code -- timed on this host over a sweep of thread counts: the best and the
single-thread figure), at N > 1 a `strong` entry (the SAME global batch of 1024 sub-trajectories
split over the ranks) next to the weak-scaling headline, and at N = 1 `secondary` entries for the
other BASELINE configurations (PPO 64 x 128 HalfCheetah shapes with the MLP and with the reference's
default LSTM policy -- the latter also at 1024 x 128 --, DDPG batch 512 off a 1e6-row uniform replay, PPO
on 256 actors of 3x84x84 camera frames), the whole on-device loops (rollout -> windows -> FIFO -> learn,
low-dim and pixel) and the host-fed learner (batches from host memory through the pinned double-buffered
ingest, incl. the path that starts from the collector's per-step Python objects).
"""
import argparse
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_T0 = time.time()            # the run's wall clock starts at import (--budget-s counts from here)

from surreal_amd import synthetic  # noqa: E402
from surreal_amd.main.ppo_configs import ppo_learner_config, ppo_env_config, ppo_session_config  # noqa: E402

B, N, D, A = 1024, 128, 376, 17
HIDDEN = (300, 200)
METRIC = 'env-steps/sec (learner ingest) PPO 1024 actors×128 steps, 1/2/4/8 MI355X'
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md (dense FP32 MFMA)
PEAK_HBM_GBPS = 8000.0
# algorithmic work of one learn() at this configuration
STEP_FLOPS = 2.0 * B * (N + 1) * (D * 300 + 300 * 200 + 200) + 10 * (3 * 2.0 * B * (D * 300 + 60000 + 200 * A) + 2.0 * B * (D * 300 + 60000 + 200 * A)) + 10 * 3 * 2.0 * B * (D * 300 + 60000 + 200) + 2.0 * B * (D * 300 + 60000 + 200 * A)
STEP_BYTES = 200466432.0


def algorithmic_costs(split_tail):
    """per-launch figures of the fused critic kernel.  With the tail
    split the launch covers the B*N step rows (the B obs_next rows go through the layered
    kernels): 1 508 B and 346 kFLOP per row either way."""
    rows = B * N if split_tail else B * (N + 1)
    flops = 2.0 * rows * (D * HIDDEN[0] + HIDDEN[0] * HIDDEN[1] + HIDDEN[1] * 1)
    bytes_ = 4.0 * (rows * D + rows)          # read obs||obs_next once, write values
    return rows, flops, bytes_


def build_learner(mode, device_index, B=B, use_graph=True):
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
    sc = ppo_session_config('/tmp/surreal_amd_bench')
    sc.learner.use_hip_graph = bool(use_graph)
    for kv in filter(None, os.environ.get('SMX_BENCH_LEARNER_OPTS', '').split(',')):     # A/B runs: key=0|1[,key=...]
        k, v = kv.split('=')
        sc.learner[k] = bool(int(v))
    learner = PPOLearner(lc, ppo_env_config(D, A), sc)
    params = synthetic.make_ppo_params(D, A, hidden=HIDDEN, seed=1)
    zstate = synthetic.make_zfilter_state(D, seed=2)
    learner.model.load_params(params)
    learner.ref_target_model.load_params(params)
    learner.model.z_filter.load_state_dict(zstate)
    learner.ref_target_model.z_filter.load_state_dict(zstate)
    return learner, params, zstate


def device_batch(learner, rank, B=B, lo=0):
    """synthetic batch of the BASELINE shape, generated once and left resident in HBM (B rows starting
    at row `lo` of the seed's 1024: a strong-scaling rank's share of the one global batch)"""
    batch = synthetic.make_ppo_batch(1024 if lo or B < 1024 else B, N, D, A, seed=100 + (0 if lo or B < 1024 else rank))
    if lo or B < 1024:
        batch = slice_batch(batch, lo, lo + B)
    return learner._preprocess_batch_ppo(copy.deepcopy(batch)), batch


def slice_batch(batch, lo, hi):
    def cut(x):
        if isinstance(x, dict):
            return type(x)((k, cut(v)) for k, v in x.items())
        if isinstance(x, list):
            return [cut(v) for v in x]
        return None if x is None else x[lo:hi]
    return cut(batch)


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
            if (('fused' in k and k.endswith('@grid%d' % (rows // 128 * 256))) or
                    ('rows16' in k and k.endswith('@grid%d' % (rows // 128 * 512)))) and 'mfma_util_pct' in v:
                return v['mfma_util_pct'], os.path.basename(path)
    return None, None


def host_cpu():
    """model name, sockets x cores (physical) and logical CPUs of this host, from lscpu"""
    import subprocess
    info = {}
    try:
        for ln in subprocess.run(['lscpu'], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if ':' in ln:
                k, v = ln.split(':', 1)
                info[k.strip()] = v.strip()
    except Exception:
        pass
    logical = os.cpu_count() or 1
    try:
        physical = int(info.get('Socket(s)', '1')) * int(info.get('Core(s) per socket', str(logical)))
    except ValueError:
        physical = logical
    return {'model': info.get('Model name', 'unknown'), 'physical_cores': min(physical, logical),
            'logical_cpus': logical}


def reference_learn_fn(params, zstate, batch, Bs, Ns, Ds, As, hyper, pixel=None):
    """a callable that runs ONE learn of the REFERENCE'S OWN learner (surreal/learner/ppo.py `_preprocess_batch_ppo` +
    `_optimize`) on `batch`, or None.  Opt-in (SMX_BENCH_REFERENCE=1) and build-container only: the reference tree does
    not travel to the GPU box in any form, so there this is always None and cpu_baseline.kind is "port" (the restatement,
    which oracle/gen_golden.py asserts bit-identical to the reference and oracle/time_reference_vs_port.py times beside
    it in the build container: profiles/r04_cpu_reference_vs_port.json)."""
    if os.environ.get('SMX_BENCH_REFERENCE') != '1':
        return None
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    try:
        import ref_shims
        if not ref_shims.reference_available():
            return None
        ref = ref_shims.import_reference()
        import gen_golden as G
        Lr = G.build_reference_learner(ref, params, zstate, Bs, Ns, Ds, As, dict(hyper, n_step=Ns), pixel=pixel)
    except Exception as e:
        sys.stderr.write('[bench.py] the reference learner could not be built: %r\n' % (e,))
        return None

    def learn():
        bd = ref_shims.BeneDict(copy.deepcopy(batch))
        bd = Lr._preprocess_batch_ppo(bd)
        Lr._optimize(bd.obs, bd.actions, bd.rewards, bd.obs_next, bd.persistent_infos, bd.onetime_infos, bd.dones)
    return learn


def _timed_learns(fn, budget_s, at_least=2, at_most=10):
    fn()                                    # warm-up
    t0, times = time.time(), []
    while len(times) < at_least or (time.time() - t0 < budget_s and len(times) < at_most):
        t1 = time.time()
        fn()
        times.append(time.time() - t1)
    return sum(times) / len(times), len(times)


def cpu_baseline(mode, params, zstate, batch, budget_s=30.0):
    """the reference learner's CPU path (oracle restatement, same ATen ops) on this host, swept over
    torch.set_num_threads (n = physical cores AND n = 1; more threads than the GEMMs
    of a 1024-row epoch can use make it slower, so the best of the sweep is the baseline)"""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ppo_oracle
    cpu = host_cpu()
    prev = torch.get_num_threads()
    counts = sorted({n for n in (1, 8, 16, 32, 64, cpu['physical_cores']) if 1 <= n <= cpu['logical_cpus']})
    O = ppo_oracle.OraclePPOLearner(params, A, B, zstate=zstate, n_step=N, ppo_mode=mode, kl_target=1e9)
    sweep = []
    per = budget_s / (len(counts) + 1)
    for n in counts:
        torch.set_num_threads(n)
        O.learn(copy.deepcopy(batch))        # warm-up at this thread count
        t0, k, times = time.time(), 0, []
        while k < 2 or (time.time() - t0 < per and k < 20):      # at least two timed learns per thread count
            t1 = time.time()
            O.learn(copy.deepcopy(batch))
            times.append(time.time() - t1)
            k += 1
        dt = (time.time() - t0) / k
        sweep.append({'threads': n, 's_per_learn': dt, 'env_steps_per_s': B * N / dt, 'learns': k,
                      's_per_learn_min': min(times), 's_per_learn_max': max(times)})
    best = max(sweep, key=lambda r: r['env_steps_per_s'])
    one = [r for r in sweep if r['threads'] == 1][0]
    port = {'value': best['env_steps_per_s'], 'cores': best['threads'], 's_per_learn': best['s_per_learn']}
    out = {'value': port['value'], 'unit': 'env-steps/s', 'cores': port['cores'], 'kind': 'port',
           'sample': 'learn() on the full 1024x128x376 batch, 10+10 epochs, %s mode, torch %s CPU, 1 warm-up + >= 2 timed '
                     'learns per thread count, best of the sweep' % (mode, torch.__version__),
           'single_thread': one['env_steps_per_s'], 'sweep': sweep, 'cpu_model': cpu['model'],
           'physical_cores': cpu['physical_cores'], 'logical_cpus': cpu['logical_cpus'], 'port': port}
    # the reference's own code (build container, opt-in) at the port's best thread count: the baseline quoted is the FASTER of the two
    fn = reference_learn_fn(params, zstate, batch, B, N, D, A, dict(ppo_mode=mode, kl_target=1e9))
    if fn is not None:
        torch.set_num_threads(best['threads'])
        dt, k = _timed_learns(fn, per)
        out['reference'] = {'value': B * N / dt, 'cores': best['threads'], 's_per_learn': dt, 'learns': k}
        if out['reference']['value'] >= port['value']:
            out.update(value=out['reference']['value'], kind='reference')
    torch.set_num_threads(prev)
    return out


# ---- the other BASELINE configurations (N = 1 only; a few steps each) -----------------------------
def _time_learn(L, db, steps, warm=2, repeats=3):
    """seconds per learn() of a secondary configuration: `repeats` timed runs of `steps` learns each, the MEDIAN run (a
    millisecond-scale learn timed once over five steps moved by 5 x between two bench runs of the same tree: one host
    hiccup of 10 ms inside the window; the headline keeps the contract's single window of exactly K steps)"""
    for _ in range(warm):
        L.learn(db)
    runs = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            L.learn(db)
        torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0) / steps)
    return sorted(runs)[len(runs) // 2]


# ---- pricing of a configuration: SURVEY.md 8(d)'s accounting, generalised over the policy ----------------
CNN_FLOPS_PER_FRAME = 2.0 * (20 * 20 * 16 * (8 * 8 * 3) + 9 * 9 * 32 * (4 * 4 * 16) + 2592 * 256)   # builders.py:8-33 on 3x84x84


def ppo_costs(Bs, Ns, Ds, As, rnn=False, pixel=None, hidden=HIDDEN, F=100, H=5, Ep=10, Ev=10):
    """ALGORITHMIC flops and bytes of one learn(): the critic over all B (N + 1) steps once, the
    policy epochs at 4 forward-equivalents each (forward + backward = 3, + the KL forward after the update,
    ppo.py:553), the value epochs at 3, the reference policy once; with an LSTM / CNN stem every pass includes the
    stem and the epochs run over B E rows (E = N - H + 1).  Bytes: what must be touched once -- observations
    (uint8 frames stay uint8), rewards, dones, the first E steps' actions and behaviour pds, values, advantages, returns."""
    E = Ns - H + 1 if rnn else 1
    Din = Ds + (256 if pixel else 0)
    stem = (2.0 * 4 * F * (Din + F) if rnn else 0.0) + (CNN_FLOPS_PER_FRAME if pixel else 0.0)     # per row-step
    top = F if rnn else Din
    Pa = 2.0 * (top * hidden[0] + hidden[0] * hidden[1] + hidden[1] * As)
    Pc = 2.0 * (top * hidden[0] + hidden[0] * hidden[1] + hidden[1] * 1)
    rows, steps_all = Bs * E, Bs * (Ns + 1)
    flops = steps_all * (stem + Pc) + Ep * 4 * rows * (stem + Pa) + Ev * 3 * rows * (stem + Pc) + rows * (stem + Pa)
    frame = (pixel[0] * pixel[1] * pixel[2]) if pixel else 0
    bytes_ = steps_all * (4.0 * Ds + frame) + 2 * 4.0 * Bs * Ns + rows * 3 * As * 4.0 + 4.0 * steps_all + 2 * 4.0 * rows + \
        (2 * 4.0 * Bs * F if rnn else 0.0)
    return flops, bytes_, dict(stem_flops_per_row_step=stem, actor_flops_per_row=Pa, critic_flops_per_row=Pc,
                               rows_per_epoch=rows, steps_all=steps_all)


def kernel_table(fn):
    """device time by kernel over ONE call of fn (torch.profiler / roctracer: every HIP kernel of the process, the
    C-ABI's included, graph replays too): [(name, calls, total_us)], largest first"""
    from torch.profiler import profile, ProfilerActivity
    import warnings
    torch.cuda.synchronize()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
    rows = [(e.key, int(e.count), float(e.device_time_total)) for e in prof.key_averages() if e.device_time_total > 0]
    return sorted(rows, key=lambda r: -r[2])


def _short(name):
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:64]


def dominant_kernel(table, family_flops):
    """the kernel family with the most device time and, where its algorithmic flops per learn are known
    (family_flops: {name prefix: flops}), its fraction of the FP32 peak (MFMA and vector rate are both 157.3 TFLOP/s)"""
    fam = {}
    for name, calls, us in table:
        k = _short(name)
        k = k.split('<')[0]
        if k.startswith('gemm'):
            k = 'gemm kernels (gemm32 / gemm_rows / gemm_tile: every dense layer)'
        elif k.startswith('lstm_fwd') or k.startswith('lstm_bwd'):
            k = k[:8] + ' kernels'
        c, t = fam.get(k, (0, 0.0))
        fam[k] = (c + calls, t + us)
    total = sum(t for _, t in fam.values())
    k, (c, t) = max(fam.items(), key=lambda kv: kv[1][1])
    out = {'name': k, 'launches_per_learn': c, 'avg_us': t / c, 'share_of_device_time': t / total,
           'device_us_per_learn': total, 'timing': 'torch.profiler (roctracer) over one learn() of this run'}
    for prefix, fl in family_flops.items():
        if k.startswith(prefix):
            out.update(algorithmic_flops_per_learn=fl, achieved_TFLOPs=fl / (t * 1e-6) / 1e12,
                       frac=fl / (t * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS)
    out['top5'] = [{'name': kk, 'launches': cc, 'us': tt} for kk, (cc, tt) in
                   sorted(fam.items(), key=lambda kv: -kv[1][1])[:5]]
    return out


def priced(dt, flops, bytes_, units, unit_name):
    """whole-learn figures against both roofs"""
    return {'algorithmic_flops_per_learn': flops, 'algorithmic_bytes_per_learn': bytes_,
            'bound': 'mfma' if flops / max(bytes_, 1.0) > PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBPS * 1e9) else 'hbm',
            'achieved_TFLOPs': flops / dt / 1e12, 'frac_of_fp32_peak': flops / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            'achieved_hbm_GBps': bytes_ / dt / 1e9, 'frac_of_hbm_peak': bytes_ / dt / 1e9 / PEAK_HBM_GBPS,
            unit_name + '_per_s': units / dt}


_CPU_BUDGET = [12.0]        # seconds the next secondary's CPU leg may take (set by secondaries())
_CPU_THREADS = [None]      # the best thread count of the headline's sweep (cpu_baseline); else the physical cores, <= 32


def _cpu_threads():
    return _CPU_THREADS[0] or max(1, min(32, host_cpu()['physical_cores']))


def cpu_ppo(Bs, Ns, Ds, As, rnn, pixel, params, batch, budget_s=None, sample_rows=None, mode='adapt'):
    """the CPU restatement of the reference learner (oracle/ppo_oracle.py, kind "port") on this host for the same
    configuration: one warm-up + timed learns inside the budget (at least one); sample_rows: a row subset of the batch
    for configurations whose full learn takes minutes on a CPU (the rate is per env-step)"""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import ppo_oracle
    budget_s = _CPU_BUDGET[0] if budget_s is None else budget_s
    prev = torch.get_num_threads()
    sampled = bool(sample_rows and sample_rows < Bs)
    if sampled:
        batch = slice_batch(batch, 0, sample_rows)
        Bs = sample_rows
    kw = dict(n_step=Ns, kl_target=1e9, ppo_mode=mode)
    if rnn:
        kw.update(if_rnn_policy=True, horizon=5)
    best = None
    t_start = time.time()
    # the headline's best thread count and 8 (small GEMMs get slower with more threads): the better of the two
    for n in sorted({8, _cpu_threads()}):
        if best is not None and time.time() - t_start > budget_s / 2:      # the first thread count used the budget up
            break
        torch.set_num_threads(n)
        O = ppo_oracle.OraclePPOLearner(params, As, Bs, **kw)
        t0 = time.time()
        O.learn(copy.deepcopy(batch))
        first = time.time() - t0
        cold = first > budget_s / 4              # a learn that takes seconds: the first call is the sample
        times = [first] if cold else []
        while not cold and len(times) < 5 and (not times or sum(times) + first < budget_s / 2):
            t1 = time.time()
            O.learn(copy.deepcopy(batch))
            times.append(time.time() - t1)
        dt = sum(times) / len(times)
        if best is None or dt < best[0]:
            best = (dt, n, len(times), cold)
    dt, n, k, cold = best
    out = {'value': Bs * Ns / dt, 'unit': 'env-steps/s', 'cores': n, 'kind': 'port', 's_per_learn': dt,
           'sample': '%d timed learn(s) of %d x %d (%s), torch %s CPU, the better of 8 and %d threads%s' % (
               k, Bs, Ns, 'full batch' if not sampled else 'the first %d sub-trajectories of the batch' % Bs,
               torch.__version__, _cpu_threads(), ' (single call, no warm-up: one learn takes seconds)' if cold else '')}
    if sampled:
        out['sample_rows'] = Bs
    out['port'] = {'value': out['value'], 'cores': n, 's_per_learn': dt}
    # the reference's own code at the same thread count (build container, opt-in) when the budget allows one more leg:
    # the baseline quoted is the faster of the two (the port is ~20 % slower than the reference on the LSTM policy)
    if time.time() - t_start < budget_s:
        fn = reference_learn_fn(params, None, batch, Bs, Ns, Ds, As, kw, pixel=pixel)
        if fn is not None:
            try:
                torch.set_num_threads(n)
                if cold:
                    t1 = time.time()
                    fn()
                    rdt, rk = time.time() - t1, 1
                else:
                    rdt, rk = _timed_learns(fn, max(0.0, budget_s - (time.time() - t_start)) / 2, at_least=1, at_most=5)
                out['reference'] = {'value': Bs * Ns / rdt, 'cores': n, 's_per_learn': rdt, 'learns': rk}
                if out['reference']['value'] >= out['value']:
                    out.update(value=out['reference']['value'], kind='reference', s_per_learn=rdt)
            except Exception as e:
                out['reference'] = {'error': repr(e)}
    torch.set_num_threads(prev)
    return out


def secondary_ppo(Bs, Ns, Ds, As, rnn, pixel=None, steps=5, cpu=True, cpu_sample_rows=None, mode='adapt'):
    from surreal_amd.learner.ppo import PPOLearner
    lc = ppo_learner_config()
    lc.algo.n_step = Ns
    lc.algo.stride = Ns
    lc.algo.rnn.if_rnn_policy = rnn
    lc.algo.ppo_mode = mode
    lc.algo.consts.kl_target = 1e9
    lc.replay.batch_size = Bs
    L = PPOLearner(lc, ppo_env_config(Ds, As, pixel=pixel), ppo_session_config('/tmp/surreal_amd_bench2'))
    F = lc.algo.rnn.rnn_hidden
    batch = synthetic.make_ppo_batch(Bs, Ns, Ds, As, seed=1, rnn_hidden=F if rnn else 0, pixel=pixel)
    db = L._preprocess_batch_ppo(copy.deepcopy(batch))
    dt = _time_learn(L, db, steps)
    flops, bytes_, parts = ppo_costs(Bs, Ns, Ds, As, rnn=rnn, pixel=pixel, F=F)
    out = {'ms_per_learn': dt * 1e3, 'env_steps_per_s': Bs * Ns / dt, 'B': Bs, 'n_step': Ns, 'obs_dim': Ds,
           'action_dim': As, 'policy': ('cnn+' if pixel else '') + ('lstm100(H=5)+mlp' if rnn else 'mlp'),
           'mode': mode, 'epochs': '10+10, KL early exit disabled', 'roofline': priced(dt, flops, bytes_, Bs * Ns, 'env_steps')}
    try:
        rows, steps_all = parts['rows_per_epoch'], parts['steps_all']
        rec = 2.0 * 4 * F * F if rnn else 0.0               # the recurrent product of one row-step
        c1, c2 = (2.0 * 20 * 20 * 16 * 192, 2.0 * 9 * 9 * 32 * 256) if pixel else (0.0, 0.0)   # the two convolutions, per frame
        n_fwd = steps_all + 22 * rows                       # row-steps through the stem forward: critic pass + ref + 11 + 10
        H1, H2 = HIDDEN
        fam = {'lstm_fwd': rec * n_fwd, 'lstm_bwd': rec * 20 * rows,
               'conv_u8_fwd': c1 * n_fwd, 'conv_cl_fwd': c2 * n_fwd, 'conv_u8_wgrad': c1 * 20 * rows,
               'conv_cl_wgrad': c2 * 20 * rows, 'conv_cl_dgrad': c2 * 20 * rows}
        if not (rnn or pixel):
            fam['mlp3_rows16_kernel'] = steps_all * parts['critic_flops_per_row']
            fam['epoch_fb_kernel'] = 10.0 * rows * (parts['actor_flops_per_row'] + parts['critic_flops_per_row'] +
                                                    2.0 * (2 * H1 * H2 + H2 * As + H2))
            fam['gemm'] = 10.0 * rows * (parts['actor_flops_per_row'] + parts['critic_flops_per_row'])   # weight gradients
        else:
            fam['gemm'] = flops - sum(fam.values())         # every dense layer (stem input products, the MLPs, the CNN's Linear)
        graph, L.use_graph = L.use_graph, False        # (kernels inside a hipGraph replay are not traced: one eager learn)
        out['dominant_kernel'] = dominant_kernel(kernel_table(lambda: L.learn(db)), fam)
        L.use_graph = graph
    except Exception as e:
        out['dominant_kernel'] = {'error': repr(e)}
    if cpu:
        try:
            params = L.model.numpy_params()            # the same (randomly initialised) parameters on both sides
            out['cpu_baseline'] = cpu_ppo(Bs, Ns, Ds, As, rnn, pixel, params, batch, sample_rows=cpu_sample_rows, mode=mode)
            out['gpu_over_cpu'] = out['env_steps_per_s'] / out['cpu_baseline']['value']
        except Exception as e:
            out['cpu_baseline'] = {'error': repr(e)}
    return out


def secondary_ddpg(steps=200, cpu=True):
    from surreal_amd.learner.ddpg import DDPGLearner
    from surreal_amd.main.ddpg_configs import ddpg_learner_config, ddpg_env_config, ddpg_session_config
    from surreal_amd.replay import UniformReplay
    Bd, Dd, Ad = 512, 17, 6
    lc = ddpg_learner_config()
    lc.replay.batch_size = Bd
    lc.replay.memory_size = 1000000
    L = DDPGLearner(lc, ddpg_env_config(Dd, Ad), ddpg_session_config())
    R = UniformReplay(lc, ddpg_env_config(Dd, Ad), ddpg_session_config())
    g = torch.Generator(device='cuda').manual_seed(0)
    for _ in range(10):
        n = 100000
        R.insert_batch({'obs': torch.randn(n, Dd, device='cuda', generator=g),
                        'obs_next': torch.randn(n, Dd, device='cuda', generator=g),
                        'actions': torch.rand(n, Ad, device='cuda', generator=g) * 2 - 1,
                        'rewards': torch.randn(n, device='cuda', generator=g),
                        'dones': (torch.rand(n, device='cuda', generator=g) < 0.01).float()})

    def sample_and_learn():
        f = R.sample_batch(Bd, out=L.staging_fields(Bd))      # gathered where the captured iteration reads its batch
        return L.learn({'obs': {'low_dim': {'flat_inputs': f['obs']}},
                        'obs_next': {'low_dim': {'flat_inputs': f['obs_next']}}, 'actions': f['actions'],
                        'rewards': f['rewards'].view(Bd, 1), 'dones': f['dones'].view(Bd, 1)})
    for _ in range(20):
        sample_and_learn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sample_and_learn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # ALGORITHMIC work of one iteration (ddpg.py:244-352): target actor + target critic forward, critic forward +
    # backward (3), actor forward, critic forward + its data gradient for the actor loss (2), actor backward (2);
    # bytes: the sampled rows + parameters, gradients and both Adam moments read and written once
    Pa = Dd * 300 + 300 * 200 + 200 * Ad
    Pc = Dd * 400 + (400 + Ad) * 300 + 300
    flops = 2.0 * Bd * (4 * Pa + 6 * Pc)
    n_par = Pa + 300 + 200 + Ad + Pc + 400 + 300 + 1
    bytes_ = 4.0 * Bd * (2 * Dd + Ad + 2) + 7 * 4.0 * n_par + 2 * 4.0 * n_par
    out = {'ms_per_iteration': dt * 1e3, 'samples_per_s': Bd / dt, 'batch': Bd, 'replay_rows': 1000000,
           'what': 'uniform sample of 512 out of 1e6 device-resident rows (into the learner\'s staging buffers) + DDPGLearner.learn',
           'roofline': priced(dt, flops, bytes_, Bd, 'samples')}
    out['roofline']['note'] = '~22 dependent launches of 512-row problems: launch-latency-bound, neither roof applies'
    try:
        graph, L.use_graph = getattr(L, 'use_graph', False), False
        out['dominant_kernel'] = dominant_kernel(kernel_table(sample_and_learn), {'gemm': flops})
        L.use_graph = graph
    except Exception as e:
        out['dominant_kernel'] = {'error': repr(e)}
    if cpu:
        try:
            sys.path.insert(0, os.path.join(ROOT, 'oracle'))
            import ddpg_oracle
            n = _cpu_threads()
            prev = torch.get_num_threads()
            torch.set_num_threads(n)
            O = ddpg_oracle.OracleDDPGLearner(ddpg_oracle.make_ddpg_params(Dd, Ad))
            b = synthetic.make_ddpg_batch(Bd, Dd, Ad, seed=0)
            for _ in range(5):
                O.learn(copy.deepcopy(b))
            t0, k = time.time(), 0
            while k < 20 or (time.time() - t0 < 3.0 and k < 2000):
                O.learn(copy.deepcopy(b))
                k += 1
            cdt = (time.time() - t0) / k
            torch.set_num_threads(prev)
            out['cpu_baseline'] = {'value': Bd / cdt, 'unit': 'samples/s', 'cores': n, 'kind': 'port', 's_per_iteration': cdt,
                                   'sample': '%d iterations of oracle/ddpg_oracle.py learn() on a resident batch of 512 '
                                             '(no replay sample on the CPU side), %d threads' % (k, n)}
            out['gpu_over_cpu'] = out['samples_per_s'] / out['cpu_baseline']['value']
        except Exception as e:
            out['cpu_baseline'] = {'error': repr(e)}
    return out


def secondary_pipeline(actors, overlap=False):
    """the whole on-device loop of one GPU (scripts/bench_pipeline.py): actors acting under the current policy on the
    synthetic environment -> window cut -> FIFO -> learn, 1024 sub-trajectories per learn"""
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import bench_pipeline
    r = bench_pipeline.run_pipeline(actors=actors, iters=5, warmup=2, graph=True, fused_step=True, overlap=overlap)
    out = {'env_steps_per_s': r['value'], 'ms_per_iteration': r['ms_per_iteration'],
           'actors': actors, 'n_step': 128, 'learns_per_rollout': r['config']['learns_per_rollout'],
           'what': 'act: ONE launch per rollout (smx_synth_rollout_f32: a workgroup owns 16 actors through all 128 steps '
                   '-- z-filter, the three policy layers on FP32 MFMA, sample, env step, record) + moving-window cut into '
                   'the FIFO table + pop + PPOLearner.learn'}
    if overlap:
        out['what'] += '; rollout k + 1 runs on a second stream while learn k runs (actors one rollout ahead, as ' \
                       'surreal\'s asynchronous agents are)'
    else:
        out.update(rollout_env_steps_per_s=r['rollout_env_steps_per_s'], stage_ms=r['stage_ms_synchronised'])
    return out


def secondary_pixel_pipeline():
    """configs[3]'s shapes as an on-device loop: 256 actors x 32 steps, every actor with a 3 x 84 x 84 uint8 camera
    rendered and stored once per step on the device, the CNN + LSTM policy acting on it (one batched act per step),
    windows (+ the LSTM state at their first step) cut on the device, FIFO, PPOLearner.learn"""
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import bench_pipeline
    r = bench_pipeline.run_pipeline(actors=256, steps=32, obs_dim=32, action_dim=8, iters=5, warmup=3, graph=False,
                                    fused_step=True, learn_batch=256, pixel=(3, 84, 84), frame_stacks=1, rnn=True)
    return {'env_steps_per_s': r['value'], 'ms_per_iteration': r['ms_per_iteration'],
            'rollout_env_steps_per_s': r['rollout_env_steps_per_s'], 'stage_ms': r['stage_ms_synchronised'],
            'actors': 256, 'n_step': 32, 'policy': 'cnn+lstm100(H=5)+mlp',
            'what': 'device camera (smx_synth_frame_u8) + frame stacking gather (smx_frame_stack_u8) + batched CNN + LSTM '
                    'act per step + env step launch; window cut incl. uint8 frames and the LSTM state at each window\'s '
                    'first step; FIFO (uint8 tables); PPOLearner.learn'}


def secondary_host_fed(iters=12):
    """batches that arrive in HOST memory (the reference's deployment: CPU agents -> collector -> replay -> learner,
    surreal/distributed/data_fetcher.py:9-73): the prefetch thread puts batch k + 1 into pinned struct-of-arrays
    staging and its host-to-device copy runs on a second stream under learn(k)
    (surreal_amd.distributed.LearnerDataPrefetcher + PinnedBatchStager).  Two producers: one that writes the pinned
    buffers in place (the rate the PCIe link allows) and one that hands over pageable arrays (+ one host memcpy)."""
    from surreal_amd.distributed import LearnerDataPrefetcher, PinnedBatchStager
    out = {}
    batch = synthetic.make_ppo_batch(B, N, D, A, seed=100)
    for name, inplace in (('producer writes the pinned staging in place', True), ('pageable host arrays (+ one host memcpy)', False)):
        learner, _, _ = build_learner('adapt', torch.cuda.current_device())
        learner.graph_input_sets = 2
        stager = PinnedBatchStager(batch, depth=2, device=learner.device)

        def filled(data, out=None):          # an in-place producer: the slot's buffers already hold the batch
            return out
        filled.accepts_out = True
        for slot in range(2):                # (both slots hold the batch once, as an in-place producer would leave them)
            stager.stage(batch, slot)
            stager.acquire(slot); stager.release(slot)
        torch.cuda.synchronize()
        pf = LearnerDataPrefetcher(learner.session_config, B, worker_preprocess=filled if inplace else None,
                                   source=lambda bs: batch, stager=stager)
        pf.start()
        for _ in range(4):
            learner.learn(pf.get())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            learner.learn(pf.get())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        pf.stop()
        out[name] = {'ms_per_batch': dt * 1e3, 'env_steps_per_s': B * N / dt,
                     'host_to_device_GBps': stager.bytes_per_batch / dt / 1e9}
        del learner, stager, pf
        torch.cuda.empty_cache()
    # the whole host path: 1024 experiences as the Python objects a collector hands over (lists of per-step arrays,
    # floats, bools) -> MultistepAggregatorWithInfo.aggregate writing straight into the pinned staging (the CPython
    # extension csrc/host/smx_host.c) in the prefetch thread -> H2D under learn(k) -> learn
    try:
        from surreal_amd.learner import aggregator as AG
        pi = batch['persistent_infos'][0]
        ob, obn = batch['obs']['low_dim']['flat_inputs'], batch['obs_next']['low_dim']['flat_inputs']
        exps = [{'obs': [{'low_dim': {'flat_inputs': ob[b, s]}} for s in range(N)],
                 'obs_next': {'low_dim': {'flat_inputs': obn[b, 0]}},
                 'actions': [batch['actions'][b, s] for s in range(N)],
                 'rewards': [float(x) for x in batch['rewards'][b]], 'dones': [bool(x) for x in batch['dones'][b]],
                 'persistent_infos': [[pi[b, s]] for s in range(N)], 'onetime_infos': [], 'n_step': N} for b in range(B)]
        learner, _, _ = build_learner('adapt', torch.cuda.current_device())
        learner.graph_input_sets = 2
        stager = PinnedBatchStager(batch, depth=2, device=learner.device)
        agg_ms = {}
        for label, native in (('extension', True), ('numpy', False)):
            saved = list(AG._NATIVE)
            if not native:
                AG._NATIVE[:] = [True, None]
            views = stager.host_views(0)
            learner.aggregator.aggregate(exps, out=views)
            t0 = time.perf_counter()
            for _ in range(3):
                learner.aggregator.aggregate(exps, out=views)
            agg_ms[label] = (time.perf_counter() - t0) / 3 * 1e3
            AG._NATIVE[:] = saved
        pf = LearnerDataPrefetcher(learner.session_config, B, worker_preprocess=learner._prefetcher_preprocess,
                                   source=lambda bs: exps, stager=stager)
        pf.start()
        for _ in range(3):
            learner.learn(pf.get())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            learner.learn(pf.get())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        pf.stop()
        out['experience dicts from CPU agents, aggregated in place by the prefetch thread'] = {
            'ms_per_batch': dt * 1e3, 'env_steps_per_s': B * N / dt, 'aggregate_ms': agg_ms,
            'native_extension': AG.native_fill() is not None}
        del learner, stager, pf
        torch.cuda.empty_cache()
    except Exception as e:
        out['experience dicts from CPU agents, aggregated in place by the prefetch thread'] = {'error': repr(e)}
    # the same with the aggregation spread over worker PROCESSES (the reference's prefetch_processes): every worker fills
    # its rows of the one staging slot in place -- shared memory, host-registered for DMA (AggregationPool)
    try:
        import functools
        from surreal_amd.distributed import SharedBatchStager, AggregationPool, PooledDataPrefetcher, ppo_aggregate_factory
        ec = ppo_env_config(D, A)
        cores = host_cpu()['physical_cores']
        for W in sorted({min(8, cores), min(16, cores), min(32, max(1, cores // 2))}):
            learner, _, _ = build_learner('adapt', torch.cuda.current_device())
            learner.graph_input_sets = 2
            example = slice_batch(batch, 0, B)
            stager = SharedBatchStager(example, depth=2, device=learner.device)
            pool = AggregationPool(stager, W, synthetic.SyntheticExperienceSource(B, N, D, A, seed0=100),
                                   functools.partial(ppo_aggregate_factory, ec.obs_spec.to_dict(), ec.action_spec.to_dict()))
            try:
                pf = PooledDataPrefetcher(learner.session_config, B, pool)
                pf.start()
                for _ in range(4):
                    learner.learn(pf.get())
                torch.cuda.synchronize()
                pf.reset_stage_times()
                t0 = time.perf_counter()
                for _ in range(10):
                    learner.learn(pf.get())
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 10
                pf.stop()
                out['experience dicts from CPU agents, aggregated in place by %d worker processes' % W] = {
                    'ms_per_batch': dt * 1e3, 'env_steps_per_s': B * N / dt, 'workers': W,
                    'slowest_worker_aggregate_ms': pool.aggregate_s * 1e3,
                    'prefetch_thread_ms_per_batch': {k: v / max(pf.stage_s['batches'], 1) * 1e3
                                                     for k, v in pf.stage_s.items() if k != 'batches'},
                    'staging': 'POSIX shared memory, hipHostRegister-ed; workers write disjoint row ranges'}
            finally:
                pool.close()
                del learner
                torch.cuda.synchronize()
                stager.close()
                torch.cuda.empty_cache()
    except Exception as e:
        import traceback
        out['experience dicts from CPU agents, aggregated in place by worker processes'] = {'error': repr(e),
                                                                                           'trace': traceback.format_exc()[-800:]}
    out['batch_bytes'] = 4 * (B * N * D + B * D + B * N * A + 2 * B * N + B * N * 2 * A)
    out['pcie_roof'] = {'GBps': 63.0, 'env_steps_per_s': B * N / (out['batch_bytes'] / 63e9)}
    out['what'] = 'pinned double-buffered staging; H2D of batch k + 1 on a copy stream under learn(k); two captured graphs ' \
                  '(one per staging slot).  The first two entries start from arrays; the third includes the host-tier ' \
                  'aggregation of 131 072 per-step Python objects per batch, which is what bounds a deployment fed by ' \
                  'remote CPU agents (per aggregating process)'
    return out


def _summary_key(key):
    for tag, short in (('configs[0]', 'cfg0 PPO 2x25 LSTM'), ('64x128, MLP', 'cfg1 PPO 64x128 MLP'),
                       ('64x128, LSTM', 'cfg1 PPO 64x128 LSTM'), ('PPO 1024x128, D=17', 'PPO 1024x128 D17 LSTM'),
                       ('configs[2]', 'cfg2 DDPG 512 of 1e6'), ('configs[3] PPO', 'cfg3 PPO 256x32 pixel CNN+LSTM'),
                       ('configs[4] clip', 'cfg4 clip MLP'), ('configs[4] LSTM, adapt', 'cfg4 adapt LSTM'),
                       ('configs[4] LSTM, clip', 'cfg4 clip LSTM'),
                       ('configs[3] on-device', 'loop 256x32 pixel'), ('4096 actors', 'loop 4096x128'),
                       ('one rollout ahead', 'loop 1024x128 overlapped'), ('on-device loop, 1024', 'loop 1024x128'),
                       ('host-fed', 'host-fed')):
        if tag in key:
            return short
    return key[:40]


def secondaries(which='core', deadline=None, cpu=True):
    """the other BASELINE configurations, most important first.  `deadline` (time.time() value): an entry that would
    start after it is skipped and says so; the CPU legs get what is left, split evenly over the entries still to run
    (at least one timed learn each -- a CPU learn of the largest shapes runs on a row sample)."""
    core = [
        # BASELINE.md section 2's cfg-5 mode matrix: {clip, adapt} x {MLP, LSTM(100) H=5} at D=376, A=17 (adapt x MLP is the headline)
        ('configs[4] clip mode, MLP policy: 1024x128x376, A=17',
         lambda c: secondary_ppo(B, N, D, A, False, steps=5, cpu=c, mode='clip', cpu_sample_rows=256)),
        ('configs[4] LSTM, adapt: 1024x128x376, A=17, LSTM(100) H=5 + MLP',
         lambda c: secondary_ppo(B, N, D, A, True, steps=3, cpu=c, cpu_sample_rows=64)),
        ('configs[4] LSTM, clip: 1024x128x376, A=17, LSTM(100) H=5 + MLP',
         lambda c: secondary_ppo(B, N, D, A, True, steps=3, cpu=c, mode='clip', cpu_sample_rows=64)),
        ('configs[1] PPO HalfCheetah shapes 64x128, LSTM policy (reference default)',
         lambda c: secondary_ppo(64, 128, 17, 6, True, cpu=c)),
        ('configs[1] PPO HalfCheetah shapes 64x128, MLP policy', lambda c: secondary_ppo(64, 128, 17, 6, False, cpu=c)),
        ('configs[2] DDPG HalfCheetah shapes, uniform replay 1e6, batch 512', lambda c: secondary_ddpg(cpu=c)),
        ('configs[3] PPO 256 actors x 32 steps, 3x84x84 uint8 frames + 32-d state, CNN + LSTM policy',
         lambda c: secondary_ppo(256, 32, 32, 8, True, pixel=(3, 84, 84), steps=3, cpu=c, cpu_sample_rows=32)),
        ('configs[0] PPO 2x25 D=17 A=6, LSTM policy (the shape of the reference test_ppo_gym --unit-test)',
         lambda c: secondary_ppo(2, 25, 17, 6, True, steps=10, cpu=c)),
        ('PPO 1024x128, D=17, A=6, LSTM policy (the reference default at the benchmark batch)',
         lambda c: secondary_ppo(1024, 128, 17, 6, True, steps=3, cpu=c, cpu_sample_rows=128)),
        ('on-device loop, 1024 actors x 128 steps (act + env step + windows + FIFO + learn)',
         lambda c: secondary_pipeline(1024)),
    ]
    extra = [
        ('on-device loop, 4096 actors x 128 steps feeding 4 learns per rollout', lambda c: secondary_pipeline(4096)),
        ('on-device loop, 1024 actors x 128 steps, actors one rollout ahead of the learner (two streams)',
         lambda c: secondary_pipeline(1024, overlap=True)),
        ('configs[3] on-device loop: 256 actors x 32 steps, 3x84x84 uint8 camera + 32-d state, CNN + LSTM policy',
         lambda c: secondary_pixel_pipeline()),
        ('host-fed learner: 1024 x 128 x 376 batches from host memory (pinned double-buffered ingest)',
         lambda c: secondary_host_fed()),
    ]
    todo = core + (extra if which == 'all' else [])
    out = {}
    for i, (key, fn) in enumerate(todo):
        now = time.time()
        if deadline is not None and now > deadline:
            out[key] = {'skipped': 'time budget (--budget-s) spent before this entry'}
            continue
        # the CPU leg's share of what is left: even split over the entries still to come, >= 1.5 s, <= 12 s
        _CPU_BUDGET[0] = 12.0 if deadline is None else min(12.0, max(1.5, 0.6 * (deadline - now) / (len(todo) - i)))
        try:
            out[key] = fn(cpu)
        except Exception as e:       # a secondary must never take the headline line down
            out[key] = {'error': repr(e)}
        if isinstance(out[key], dict):
            out[key]['wall_s'] = time.time() - now
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


# ---- the ONE stdout line: <= 4 KB (the driver keeps an 8 KB tail); everything else goes to --full-out -------------
LINE_LIMIT = 4096


def _r(x, sig=5):
    """floats to `sig` significant digits (the full-precision figures are in the --full-out record)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float('%.*g' % (sig, x))
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def summary_row(r):
    """<= 6 keys per secondary configuration: rate | ms | whole-learn fraction of its bounding roof | the dominant
    kernel family's own fraction | the CPU baseline (value@cores, kind) | GPU / CPU"""
    if not isinstance(r, dict):
        return None
    if 'error' in r or 'skipped' in r:
        return {'error': str(r.get('error', r.get('skipped')))[:80]}
    row = {'rate': r.get('env_steps_per_s', r.get('samples_per_s')), 'ms': r.get('ms_per_learn', r.get('ms_per_iteration'))}
    if row['rate'] is None:          # an entry of sub-entries (the host-fed learner's producers): the best of them
        subs = [v for v in r.values() if isinstance(v, dict) and 'env_steps_per_s' in v]
        if subs:
            best = max(subs, key=lambda v: v['env_steps_per_s'])
            row = {'rate': best['env_steps_per_s'], 'ms': best.get('ms_per_batch')}
    rf = r.get('roofline')
    if isinstance(rf, dict):
        row['frac_' + str(rf.get('bound', 'mfma'))] = rf.get('frac_of_fp32_peak') if rf.get('bound') == 'mfma' else rf.get('frac_of_hbm_peak')
    dk = r.get('dominant_kernel')
    if isinstance(dk, dict) and dk.get('frac') is not None:
        row['dom_frac'] = dk['frac']
    cb = r.get('cpu_baseline')
    if isinstance(cb, dict) and cb.get('value') is not None:
        row['cpu'] = '%.4g@%d %s' % (cb['value'], cb.get('cores', 0), cb.get('kind', 'port'))
        if cb.get('sample_rows'):       # the CPU learn ran on a row sample of the batch: its rate is per env-step of the SAMPLE
            row['cpu'] += ' extrapolated from %d rows' % cb['sample_rows']
        row['x_cpu'] = r.get('gpu_over_cpu')
    return {k: v for k, v in row.items() if v is not None}


def compact_line(full, full_path=None):
    """the driver-facing line from the full record: the contract's keys, `roofline`, `step_roofline`, `cpu_baseline`,
    `secondary_summary`; shrunk further (summary first, then optional keys) if it would still exceed LINE_LIMIT"""
    out = _pick(full, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling'))
    for k in ('value', 'ms_per_step'):                 # null is meaningful here (diagnostic line)
        out.setdefault(k, full.get(k))
    out['vs_baseline'] = full.get('vs_baseline')
    out.update(_pick(full, ('dtype', 'data', 'error')))
    cfg = full.get('config', {})
    out['config'] = _pick(cfg, ('workload', 'mode', 'B_per_gpu', 'n_step', 'obs_dim', 'action_dim', 'hidden', 'epochs', 'hip_graph',
                                'parallelism', 'epoch_kernels', 'collectives_per_step', 'exchange', 'exchange_fallback_reason',
                                'rccl_ranks', 'epoch_all_reduce_us', 'epoch_all_reduce_bytes', 'graph_segments'))
    if isinstance(out['config'].get('exchange_fallback_reason'), str):
        out['config']['exchange_fallback_reason'] = out['config']['exchange_fallback_reason'][:160]
    if 'roofline' in full:
        out['roofline'] = _pick(full['roofline'], ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source',
                                                   'traffic_measured_in_run', 'mfma_busy_pct', 'mfma_busy_measured_in_run',
                                                   'kernel_ms', 'algorithmic_bytes_per_launch',
                                                   'flops_per_launch', 'share_of_step'))
        out['roofline'].setdefault('traffic', None)
    if 'step_roofline' in full:
        out['step_roofline'] = _pick(full['step_roofline'], ('mfma_frac', 'hbm_frac', 'algorithmic_flops_per_learn',
                                                             'algorithmic_bytes_per_learn'))
    if 'cpu_baseline' in full:
        out['cpu_baseline'] = _pick(full['cpu_baseline'], ('value', 'unit', 'cores', 'kind', 'sample', 'single_thread', 'cpu_model',
                                                           'physical_cores', 'port', 'reference'))
        out.update(_pick(full, ('gpu_over_cpu', 'gpu_over_cpu_single_thread')))
    if 'strong' in full:
        out['strong'] = _pick(full['strong'], ('value', 'unit', 'ms_per_step', 'global_batch', 'B_per_gpu', 'hip_graph'))
    if 'exchange_model' in full:
        out['exchange_model'] = _pick(full['exchange_model'], ('exchange_ms_per_step', 'share_of_step', 'predicted_efficiency'))
    if 'secondary' in full:
        out['secondary_summary'] = {_summary_key(k): summary_row(r) for k, r in full['secondary'].items()
                                    if summary_row(r) is not None}
    out.update(_pick(full, ('final_stats', 'bench_wall_s')))
    if full_path:
        out['full_record'] = full_path
    out = _r(out)
    line = json.dumps(out)
    # belt and braces: never hand the driver a line it cannot keep whole
    # (the strong-scaling row and the exchange fields are what an N > 1 line exists for: they go last)
    for drop in ('final_stats', 'exchange_model', 'step_roofline'):
        if len(line) <= LINE_LIMIT:
            break
        out.pop(drop, None)
        line = json.dumps(out)
    while len(line) > LINE_LIMIT and out.get('secondary_summary'):
        out['secondary_summary'].popitem()
        line = json.dumps(out)
    if len(line) > LINE_LIMIT:
        out.pop('strong', None)
        line = json.dumps(out)
    if len(line) > LINE_LIMIT:
        out['config'] = _pick(out['config'], ('workload', 'parallelism'))
        out['config']['workload'] = out['config'].get('workload', '')[:120]
        line = json.dumps(out)
    return line


def write_full(full, path):
    """the complete record (every configuration's full entry) as a file; returns the path written, or None"""
    if not path:
        return None
    try:
        d = os.path.dirname(os.path.abspath(path))
        os.makedirs(d, exist_ok=True)
        with open(path, 'w') as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
    except OSError as e:
        sys.stderr.write('[bench.py] could not write %s: %r\n' % (path, e))
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--mode', default='adapt', choices=['adapt', 'clip'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='weak: 1024 sub-trajectories per GPU (the headline); strong: 1024 in total')
    ap.add_argument('--no-secondary', action='store_true', help='skip the other BASELINE configurations')
    ap.add_argument('--secondary', default='core', choices=['core', 'all'],
                    help='core: the BASELINE configurations + the 1024-actor on-device loop (fits the default budget); '
                         'all: also the 4096-actor / overlapped / pixel loops and the host-fed learner')
    ap.add_argument('--budget-s', type=float, default=70.0,
                    help='wall-clock budget of the whole run: CPU baselines shrink to what is left, secondaries that '
                         'would start after it are skipped (and say so)')
    ap.add_argument('--full-out', default=os.path.join(ROOT, 'gpurun_out', 'bench_full.json'),
                    help='the complete record (the stdout line is its <= 4 KB summary); "" to skip')
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it (the shape of the driver's N = 1 command):
    # launch the N ranks ourselves -- the same command the docstring names -- instead of refusing
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        diagnostic(args, 'WORLD_SIZE is %d but --gpus is %d: launch with torch.distributed.run --nproc-per-node %d '
                         '(or plain `python bench.py --gpus %d`, which launches the ranks itself)'
                   % (world, args.gpus, args.gpus, args.gpus), rank)
        return 2
    # SMX_BENCH_BACKEND=gloo is the single-GPU rehearsal of the N > 1 path (tests/test_gpu_dist.py):
    # RCCL refuses two ranks on one device, gloo does not, and the ranks then share the GPU
    backend = os.environ.get('SMX_BENCH_BACKEND', 'nccl')
    ndev = torch.cuda.device_count()
    if ndev == 0 or (backend == 'nccl' and world > ndev):
        diagnostic(args, 'this node shows %d GPU(s) to the process (HIP_VISIBLE_DEVICES=%r) but --gpus is %d; RCCL needs '
                         'one device per rank (SMX_BENCH_BACKEND=gloo rehearses the N > 1 path on fewer devices)'
                   % (ndev, os.environ.get('HIP_VISIBLE_DEVICES'), world), rank)
        return 0
    if backend != 'nccl':
        local_rank %= ndev
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        import datetime
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # a bounded rendezvous / collective timeout: first contact with a node must end in a line, not in a hang
        tmo = datetime.timedelta(seconds=float(os.environ.get('SMX_BENCH_PG_TIMEOUT_S', '180')))
        try:
            if backend == 'nccl':
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=tmo)
            else:
                dist.init_process_group(backend, timeout=tmo)
        except Exception as e:
            diagnostic(args, 'init_process_group(%s) failed: %r' % (backend, e), rank)
            return 0
    try:
        return run(args, world, rank, local_rank, backend)
    except Exception as e:              # the line is the contract: say what stopped the run, then fail
        import traceback
        traceback.print_exc()
        diagnostic(args, 'rank %d: %r' % (rank, e), 0)
        return 1


def _exchange_failure():
    from surreal_amd.distributed.peer_exchange import PeerExchange
    return PeerExchange.last_failure


def diagnostic(args, why, rank=0):
    """the ONE JSON line when no measurement could be made (value null + the reason)"""
    if rank == 0:
        print(compact_line({'metric': METRIC, 'value': None, 'unit': 'env-steps/s', 'n_gpus': args.gpus, 'steps': args.steps,
                            'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': args.scaling,
                            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                            'config': {'workload': 'BASELINE configs[4]', 'parallelism': 'dp%d' % args.gpus},
                            'error': str(why)[-1500:]}), flush=True)


def self_launch(args):
    """`python bench.py --gpus N`: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port <free> bench.py <same flags>` under a wall-clock budget, pass its JSON line through, and
    print a diagnostic line of our own if it ends without one"""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if os.environ.get('SMX_BENCH_BACKEND', 'nccl') == 'nccl' and ndev < args.gpus:
        diagnostic(args, 'this node shows %d GPU(s) (HIP_VISIBLE_DEVICES=%r) but --gpus is %d; RCCL needs one device per '
                         'rank' % (ndev, os.environ.get('HIP_VISIBLE_DEVICES'), args.gpus))
        return 0
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    budget = float(os.environ.get('SMX_BENCH_LAUNCH_BUDGET_S', '1500'))
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=budget, cwd=ROOT)
        out, err, rc = r.stdout, r.stderr, r.returncode
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or '')
        err = e.stderr.decode() if isinstance(e.stderr, bytes) else (e.stderr or '')
        rc = 124
        err += '\n[bench.py] the %d-rank run exceeded its wall-clock budget of %.0f s' % (args.gpus, budget)
    sys.stderr.write(err[-8000:])
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    if lines:
        print(lines[-1], flush=True)
        return rc
    diagnostic(args, 'the %d-rank launch ended (rc %d) without a result line: %s' % (args.gpus, rc, err.strip()[-600:]))
    return rc or 1


def run(args, world, rank, local_rank, backend):
    import torch.distributed as dist

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(Bl, lo):
        """W warm-up steps, then exactly K steps between barriers + device syncs; max over ranks"""
        learner, params, zstate = build_learner(args.mode, local_rank, B=Bl, use_graph=not args.no_graph)
        dbatch, batch = device_batch(learner, rank, B=Bl, lo=lo)
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
        return learner, params, zstate, dbatch, batch, dict(stats), dt

    strong_lo, strong_hi = rank * B // world, (rank + 1) * B // world
    strong = None
    if args.scaling == 'strong':
        learner, params, zstate, dbatch, batch, stats, dt = timed_run(strong_hi - strong_lo, strong_lo)
        total_rows = B
    else:
        learner, params, zstate, dbatch, batch, stats, dt = timed_run(B, 0)
        total_rows = world * B
        if world > 1:       # the same global batch of 1024 sub-trajectories split over the ranks
            sl, _, _, _, _, sstats, sdt = timed_run(strong_hi - strong_lo, strong_lo)
            strong = {'value': B * N * args.steps / sdt, 'unit': 'env-steps/s', 'ms_per_step': sdt / args.steps * 1e3,
                      'global_batch': B, 'B_per_gpu': strong_hi - strong_lo, 'hip_graph': bool(sl.use_graph),
                      'final_stats': {k: sstats[k] for k in ('_surr_loss', '_val_loss', '_pol_kl') if k in sstats}}
            if sl._dist.exchange is not None:
                barrier()
                sl._dist.exchange.close()
            del sl

    ws = learner._ws
    # what one per-epoch exchange costs on this node: the all-reduce of the epoch payload, timed alone -- through the
    # path the learner used (PeerExchange kernels over IPC-mapped peer buffers when its self-check passed, else the
    # process group) and through the process group (RCCL) for comparison
    collective_us = collective_bytes = pg_us = None
    exchange_kind = getattr(learner, 'exchange_kind', None)
    if world > 1 and getattr(ws, 'ar', None) is not None:
        def time_all_reduce(fn):
            for _ in range(5):
                fn(ws.ar)
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn(ws.ar)
            e1.record()
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / 20 * 1e3], device='cuda', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        pg_us = time_all_reduce(dist.all_reduce)
        collective_us = time_all_reduce(learner._dist.all_reduce) if learner._dist.exchange is not None else pg_us
        collective_bytes = ws.ar.numel() * 4
    kt = time_fused_kernel(learner, dbatch) if ws.key[0] == B and not (learner.if_rnn_policy or learner.model.if_pixel) else None
    out = None
    if rank == 0:
        step_s = dt / args.steps
        out = {
            'metric': METRIC,
            'value': total_rows * N * args.steps / dt,
            'unit': 'env-steps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': step_s * 1e3,
            'higher_is_better': True,
            'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': 'BASELINE configs[4]: PPO synthetic 1024 actors x 128 steps x 376-dim obs %s, one learn() '
                            'per step, batch resident in HBM' % (
                                'per GPU' if args.scaling == 'weak' else 'in total, split over the GPUs'),
                'mode': args.mode, 'hidden': list(HIDDEN), 'epochs': '10 policy + 10 value, KL early exit disabled',
                'B_per_gpu': ws.key[0], 'n_step': N, 'obs_dim': D, 'action_dim': A,
                'hip_graph': bool(learner.use_graph), 'parallelism': 'dp%d' % world,
                'epoch_kernels': 'fused row-block' if getattr(ws, 'fused', False) else 'layered',
                'collectives_per_step': getattr(learner, 'collectives_per_step', 0 if world == 1 else None),
                'exchange': exchange_kind,
                'rccl_ranks': world if world > 1 else None,      # ranks in the process group (backend nccl = RCCL; SMX_BENCH_BACKEND=gloo in rehearsals)
                'exchange_fallback_reason': _exchange_failure() if world > 1 else None,
                'epoch_all_reduce_us': collective_us, 'epoch_all_reduce_bytes': collective_bytes,
                'epoch_all_reduce_us_process_group': pg_us,
                'graph_segments': bool(world > 1 and learner.use_graph and getattr(learner._dist, 'exchange', None) is None),
                'zfilter': 'reciprocal (<= 1 ulp from z_filter.py:77; the exact division is selectable, DESIGN.md 1)',
            },
            'final_stats': {k: stats[k] for k in ('_surr_loss', '_val_loss', '_pol_kl') if k in stats},
        }
        if kt is not None:
            rows, flops, bytes_ = algorithmic_costs(ws.split_tail)
            traffic, tsrc = measured_traffic(rows)
            busy, bsrc = measured_mfma_util(rows)
            out['roofline'] = {
                'kernel': 'mlp3_rows16_kernel<19,13,true> (z-filter + critic MLP, %d rows)' % rows,
                'bound': 'mfma',
                'achieved': flops / kt / 1e12,
                'peak': PEAK_FP32_MFMA_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': flops / kt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                # PMC counters cannot be collected inside this process: these two come from the committed
                # rocprofv3 --pmc passes of the same command (profiles/), NOT from this run
                'traffic': traffic, 'traffic_source': tsrc, 'traffic_measured_in_run': False,
                'mfma_busy_pct': busy, 'mfma_busy_source': bsrc, 'mfma_busy_measured_in_run': False,
                'kernel_ms': kt * 1e3,          # measured in this run (HIP events on the launch stream)
                'flops_per_launch': flops,
                'algorithmic_bytes_per_launch': bytes_,
                'hbm_GBps': bytes_ / kt / 1e9,
                'hbm_frac': bytes_ / kt / 1e9 / PEAK_HBM_GBPS,
                'share_of_step': kt / step_s,
            }
        if args.scaling == 'weak':
            # the whole learn() against both roofs (71 GFLOP and 200 466 432 B per learn and GPU)
            out['step_roofline'] = {
                'algorithmic_flops_per_learn': STEP_FLOPS, 'algorithmic_bytes_per_learn': STEP_BYTES,
                'achieved_mfma_TFLOPs': STEP_FLOPS / step_s / 1e12,
                'mfma_frac': STEP_FLOPS / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                'achieved_hbm_GBps': STEP_BYTES / step_s / 1e9,
                'hbm_frac': STEP_BYTES / step_s / 1e9 / PEAK_HBM_GBPS,
            }
        if world > 1 and collective_us is not None:
            # the serial share of the exchanges in a step, if none of them overlapped with compute (the two large
            # ones per epoch do not: the next launch needs their result), and the weak-scaling efficiency that
            # leaves: t_1 / t_N with t_1 = t_N - exchanges
            per_step = out['config']['collectives_per_step'] or 0
            ex_ms = per_step * collective_us / 1e3
            out['exchange_model'] = {'exchange_ms_per_step': ex_ms, 'share_of_step': ex_ms / (step_s * 1e3),
                                     'predicted_efficiency': max(0.0, 1.0 - ex_ms / (step_s * 1e3)),
                                     'note': 'collectives_per_step x epoch_all_reduce_us (the small exchanges are '
                                             'priced like the large one: an upper bound)'}
        if strong is not None:
            out['strong'] = strong
        if world == 1 and args.scaling == 'weak':
            if not args.no_cpu_baseline:
                out['cpu_baseline'] = cpu_baseline(args.mode, params, zstate, batch, budget_s=min(15.0, 0.25 * args.budget_s))
                out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
                out['gpu_over_cpu_single_thread'] = out['value'] / out['cpu_baseline']['single_thread']
                _CPU_THREADS[0] = out['cpu_baseline']['cores']
            if not args.no_secondary:
                del learner, dbatch
                torch.cuda.empty_cache()
                out['secondary'] = secondaries(args.secondary, _T0 + args.budget_s, cpu=not args.no_cpu_baseline)
        out['bench_wall_s'] = time.time() - _T0
        print(compact_line(out, write_full(out, args.full_out)), flush=True)
    if world > 1:
        dist.barrier()
        if getattr(learner, '_dist', None) is not None and learner._dist.exchange is not None:
            learner._dist.exchange.close()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main() or 0)
