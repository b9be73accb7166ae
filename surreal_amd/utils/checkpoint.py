"""
Checkpoints in the reference's on-disk format (surreal/utils/checkpoint.py:17-395) -- the FORMAT is the contract
(a folder written by either side must restore on the other; tests/test_wire_formats.py holds a folder recorded
from the reference):

    <folder>/<name>.<global_steps>.ckpt       pickle of OrderedDict(attr -> state_dict() | value)
    <folder>/<name>.best-<steps>.ckpt         copies of the best-scoring ones (keep_best > 0)
    <folder>/metadata.<name>.yml              version, save_counter, global_steps, tracked_attrs,
                                              keep_history, keep_best, history_ckpt_files (newest
                                              first), best_ckpt_files, best_scores, ckpt{file: info}

Three pieces: `_Layout` knows the file names, `_Manifest` is the yml (history rotation and the best-score ranking
are list operations on it that return the files falling out), `Checkpoint` moves the tracked attributes between the
object and the pickle and keeps the folder in step with the manifest.

An attribute is stored through ``state_dict()`` / restored through ``load_state_dict()`` when it has them (the
reference tests ``isinstance(torch.nn.Module | Optimizer)``; the models here are not nn.Modules but keep the same
two methods), anything else is pickled as is.  Tensors are moved to the host before pickling, so a checkpoint does
not depend on the device it was written from.
"""
import bisect
import collections
import copy
import datetime
import os
import pickle
import shutil
import time

import torch
import yaml

CHECKPOINT_VERSION = '0.0.1'       # surreal/utils/checkpoint.py:14


def _host_copy(tree):
    """the same nested structure with every tensor on the host"""
    if torch.is_tensor(tree):
        return tree.detach().cpu()
    if isinstance(tree, dict):
        # OrderedDict stays an OrderedDict; any other dict subclass keeps its type through a shallow copy + item
        # assignment (a defaultdict's constructor wants its factory first, a Counter's would count the pairs)
        if type(tree) in (dict, collections.OrderedDict):
            return type(tree)((k, _host_copy(v)) for k, v in tree.items())
        out = copy.copy(tree)
        for k, v in tree.items():
            out[k] = _host_copy(v)
        return out
    if isinstance(tree, tuple) and hasattr(tree, '_fields'):     # namedtuple: positional constructor
        return type(tree)(*(_host_copy(v) for v in tree))
    if isinstance(tree, (list, tuple)):
        return type(tree)(_host_copy(v) for v in tree)
    return tree


def _has_state(obj):
    return callable(getattr(obj, 'state_dict', None)) and callable(getattr(obj, 'load_state_dict', None))


class _Layout(object):
    """file names of one named checkpoint series inside a folder"""

    def __init__(self, folder, name):
        self.folder, self.name = os.path.expanduser(folder), name

    def file(self, suffix):
        return '%s.%s.ckpt' % (self.name, suffix)

    def best_file(self, suffix):
        return self.file('best-%s' % (suffix,))

    def manifest(self):
        return 'metadata.%s.yml' % (self.name,)

    def at(self, fname, folder=None):
        return os.path.join(folder or self.folder, fname)

    def unlink(self, fname):
        try:
            os.remove(self.at(fname))
        except FileNotFoundError:
            pass


class _Manifest(object):
    """the yml as a dict (`.d`, what the reference calls metadata) plus the two retention rules"""

    def __init__(self, d):
        self.d = d

    @classmethod
    def fresh(cls, tracked_attrs, keep_history, keep_best):
        if tracked_attrs is not None:
            if not isinstance(tracked_attrs, (list, tuple)) or not all(isinstance(a, str) for a in tracked_attrs):
                raise AssertionError('tracked_attrs must be a list of attribute name strings or None')
            tracked_attrs = list(tracked_attrs)
        assert keep_history >= 1 and keep_best >= 0
        return cls(dict(version=CHECKPOINT_VERSION, save_counter=0, tracked_attrs=tracked_attrs,
                        keep_history=keep_history, keep_best=keep_best, history_ckpt_files=[], best_ckpt_files=[],
                        best_scores=[], ckpt={}))

    @classmethod
    def read(cls, path):
        with open(path) as fp:
            d = yaml.safe_load(fp)
        if str(d.get('version')) != CHECKPOINT_VERSION:
            raise ValueError('checkpoint version incompatible, please examine {} and make sure it is {}'
                             .format(path, CHECKPOINT_VERSION))
        return cls(d)

    def write(self, path):
        with open(path, 'w') as fp:
            yaml.safe_dump(self.d, fp, default_flow_style=False)

    def push_history(self, fname):
        """fname becomes the newest history entry -> the files that fall off the end"""
        files = [fname] + [f for f in self.d['history_ckpt_files'] if f != fname]
        keep = self.d['keep_history']
        self.d['history_ckpt_files'] = files[:keep]
        return files[keep:]

    def rank_best(self, score, fname):
        """insert (score, fname) into the best list (descending scores; a tie goes IN FRONT of the entries already
        there, checkpoint.py:357-395) -> (whether fname made it, the files that fall off)"""
        scores, files = list(self.d['best_scores']), list(self.d['best_ckpt_files'])
        keep = self.d['keep_best']
        dropped = files[keep:]
        del scores[keep:], files[keep:]
        at = bisect.bisect_left([-s for s in scores], -score)
        scores.insert(at, score)
        files.insert(at, fname)
        dropped += files[keep:]
        del scores[keep:], files[keep:]
        self.d['best_scores'], self.d['best_ckpt_files'] = scores, files
        return fname in files, dropped


class Checkpoint(object):
    def __init__(self, folder, name, *, tracked_obj, tracked_attrs=None, keep_history=1,
                 keep_best=1, mkdir=True):
        self._layout = _Layout(folder, name)
        if mkdir:
            os.makedirs(self._layout.folder, exist_ok=True)
        self.tracked_obj = tracked_obj
        if os.path.exists(self.metadata_path()):
            self._manifest = _Manifest.read(self.metadata_path())
        else:
            self._manifest = _Manifest.fresh(tracked_attrs, keep_history, keep_best)

    # ---- what callers of the reference's class read --------------------------------------------------
    @property
    def folder(self):
        return self._layout.folder

    @property
    def name(self):
        return self._layout.name

    @property
    def metadata(self):
        return self._manifest.d

    def metadata_name(self):
        return self._layout.manifest()

    def metadata_path(self):
        return self._layout.at(self._layout.manifest())

    def ckpt_name(self, suffix):
        return self._layout.file(suffix)

    def ckpt_path(self, suffix):
        return self._layout.at(self._layout.file(suffix))

    # ---- object <-> pickle ---------------------------------------------------------------------------
    def _snapshot(self):
        attrs = self.metadata['tracked_attrs']
        assert attrs is not None, 'tracked_attrs must not be None for save(). ' \
                                  'Did you forget to restore from an existing checkpoint?'
        out = collections.OrderedDict()
        for a in attrs:
            v = getattr(self.tracked_obj, a)
            out[a] = _host_copy(v.state_dict() if _has_state(v) else v)
        return out

    def _apply(self, data):
        for a in self.metadata['tracked_attrs']:
            v = getattr(self.tracked_obj, a)
            if _has_state(v):
                v.load_state_dict(data[a])
            else:
                setattr(self.tracked_obj, a, data[a])

    # ---- save ----------------------------------------------------------------------------------------
    def save(self, score=None, global_steps=None, reload_metadata=False, **ckpt_info):
        L = self._layout
        if reload_metadata:
            self._manifest = _Manifest.read(self.metadata_path())
        man, d = self._manifest, self._manifest.d
        d['save_counter'] += 1
        steps = d['save_counter'] if global_steps is None else global_steps
        fname = L.file(steps)
        with open(L.at(fname), 'wb') as fp:
            pickle.dump(self._snapshot(), fp)
        d['global_steps'] = steps
        for old in man.push_history(fname):
            L.unlink(old)
        info = dict(score=score, global_steps=steps, save_counter=d['save_counter'], time=time.time(),
                    datetime=str(datetime.datetime.now()), **ckpt_info)
        d['ckpt'][fname] = info
        if d['keep_best'] > 0:
            assert score is not None, 'score cannot be None if keep_best is enabled'
            best = L.best_file(steps)
            kept, dropped = man.rank_best(score, best)
            if kept:
                shutil.copy(L.at(fname), L.at(best))
                d['ckpt'][best] = dict(info)
            for f in dropped:
                L.unlink(f)
                d['ckpt'].pop(f, None)
        man.write(self.metadata_path())

    # ---- restore -------------------------------------------------------------------------------------
    def _manifest_of(self, folder):
        """the manifest of this series in `folder` becomes the current one"""
        self._manifest = _Manifest.read(self._layout.at(self._layout.manifest(), folder))

    def _load_file(self, fname, folder, must_exist):
        path = self._layout.at(fname, folder)
        if not os.path.exists(path):
            if must_exist:
                raise FileNotFoundError(path + ' missing.')
            return None
        with open(path, 'rb') as fp:
            self._apply(pickle.load(fp))
        return path

    def restore(self, target, mode, reload_metadata=True, check_ckpt_exists=False,
                restore_folder=None):
        """target: int n = the n-th newest (or n-th best), or the global-steps suffix of a file"""
        assert mode in ('best', 'history')
        folder = os.path.expanduser(restore_folder) if restore_folder else self.folder
        if reload_metadata or restore_folder:
            self._manifest_of(folder)
        if isinstance(target, int):
            assert target >= 0, 'target int should start from 0 for the last or best'
            ranked = self.metadata['best_ckpt_files' if mode == 'best' else 'history_ckpt_files']
            if target >= len(ranked):
                if check_ckpt_exists:
                    raise FileNotFoundError('{} [{}] ckpt file missing'.format(mode.capitalize(), target))
                return None
            fname = ranked[target]
        else:
            assert '.ckpt' not in str(target), 'use restore_full_name() instead'
            fname = self._layout.best_file(target) if mode == 'best' else self._layout.file(target)
        return self._load_file(fname, folder, check_ckpt_exists)

    def restore_full_name(self, ckpt_file, check_ckpt_exists=True, restore_folder=None):
        folder = os.path.expanduser(restore_folder) if restore_folder else self.folder
        self._manifest_of(folder)
        return self._load_file(ckpt_file, folder, check_ckpt_exists)


class PeriodicCheckpoint(Checkpoint):
    """saves on every `period`-th call, at most once per `min_interval` seconds
    (checkpoint.py:316-354)"""

    def __init__(self, *args, period, min_interval=0, **kwargs):
        super().__init__(*args, **kwargs)
        assert period >= 1
        self.period = period
        self.min_interval = min_interval
        self._calls = 0
        self.last_update_time = time.time()

    def save(self, *args, **kwargs):
        self._calls += 1
        due = self._calls % self.period == 0 and time.time() - self.last_update_time >= self.min_interval
        if due:
            super().save(*args, **kwargs)
            self.last_update_time = time.time()
        return due

    def reset_period(self):
        self._calls = 0
