"""does a co-tenant kernel on a second stream really take CUs away from learn()?  (diagnostic for
tests/test_gpu_learner.py::test_fused_fwdbwd_*)"""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch
import helpers as H
from surreal_amd import _lib as L
g, case = H.load_golden('cfg5_adapt')
batch, params, zstate = H.case_inputs(case)
for blocks, us in ((0, 0), (192, 4000), (224, 20000), (224, 450000), (240, 450000)):
    learner = H.make_learner(case, params, zstate)
    db = learner._preprocess_batch_ppo(copy.deepcopy(batch))
    learner.learn(db); torch.cuda.synchronize()           # warm-up + capture
    for m in (learner.model, learner.ref_target_model):
        m.load_params(params)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.time()
    if blocks:
        with torch.cuda.stream(side):
            learner.K.device_occupy(blocks, us)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    try:
        stats = learner.learn(db)
        ev1.record()
        torch.cuda.synchronize()
        err = int(learner._ws.ctrl_i[L.C_SYNC_ERR].item()) if learner._ws is not None else -1
        msg = 'ok'
        try:
            dict(stats)
        except RuntimeError as e:
            msg = 'raised: ' + str(e)[:60]
    except RuntimeError as e:
        torch.cuda.synchronize()
        err, msg = -2, 'raised in learn: ' + str(e)[:60]
    print('tenant %3d blocks x %6d us: learn %.2f ms (events), wall %.1f ms, sync_err %d, fb %s, %s' % (
        blocks, us, ev0.elapsed_time(ev1) if msg == 'ok' or 'raised:' in msg else -1, (time.time() - t0) * 1e3, err,
        getattr(learner._ws, 'fb', None) if learner._ws is not None else None, msg), flush=True)
