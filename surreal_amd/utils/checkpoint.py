"""
Checkpoints in the reference's on-disk format (surreal/utils/checkpoint.py:17-395):

    <folder>/<name>.<global_steps>.ckpt       pickle of OrderedDict(attr -> state_dict() | value)
    <folder>/<name>.best-<steps>.ckpt         copies of the best-scoring ones (keep_best > 0)
    <folder>/metadata.<name>.yml              version, save_counter, global_steps, tracked_attrs,
                                              keep_history, keep_best, history_ckpt_files (newest
                                              first), best_ckpt_files, best_scores, ckpt{file: info}

An attribute is stored through ``state_dict()`` / restored through ``load_state_dict()`` when it has
them (the reference tests ``isinstance(torch.nn.Module | Optimizer)``; the models here are not
nn.Modules but keep the same two methods), anything else is pickled as is.  Tensors are moved to
the host before pickling, so a checkpoint does not depend on the device it was written from.
"""
import collections
import datetime
import os
import pickle
import shutil
import time

import torch
import yaml

CHECKPOINT_VERSION = '0.0.1'       # surreal/utils/checkpoint.py:14


def _to_host(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, collections.OrderedDict):
        return collections.OrderedDict((k, _to_host(v)) for k, v in obj.items())
    if isinstance(obj, dict):
        return {k: _to_host(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_host(v) for v in obj)
    return obj


def _stateful(obj):
    return hasattr(obj, 'state_dict') and hasattr(obj, 'load_state_dict')


class _ScoreQueue(object):
    """(score, file) pairs, best first, at most max_size (checkpoint.py:357-395)"""

    def __init__(self, max_size):
        self._queue = []
        self.max_size = max_size

    def set_queue(self, scores, filepaths):
        self._queue = list(zip(scores, filepaths))
        dropped = self._queue[self.max_size:]
        del self._queue[self.max_size:]
        return dropped

    def add(self, score, filepath):
        i = len(self._queue)
        while i >= 1 and not self._queue[i - 1][0] > score:
            i -= 1
        self._queue.insert(i, (score, filepath))
        if len(self._queue) > self.max_size:
            return self._queue.pop()
        return None

    def get_scores_filepaths(self):
        if not self._queue:
            return [], []
        scores, files = zip(*self._queue)
        return list(scores), list(files)


class Checkpoint(object):
    def __init__(self, folder, name, *, tracked_obj, tracked_attrs=None, keep_history=1,
                 keep_best=1, mkdir=True):
        self.folder = os.path.expanduser(folder)
        if mkdir:
            os.makedirs(self.folder, exist_ok=True)
        self.name = name
        self.tracked_obj = tracked_obj
        if os.path.exists(self.metadata_path()):
            self._load_metadata()
        else:
            self._check_tracked_attrs(tracked_attrs)
            assert keep_history >= 1 and keep_best >= 0
            self.metadata = {
                'version': CHECKPOINT_VERSION, 'save_counter': 0, 'history_ckpt_files': [],
                'ckpt': {}, 'tracked_attrs': list(tracked_attrs) if tracked_attrs is not None else None,
                'keep_history': keep_history, 'keep_best': keep_best, 'best_ckpt_files': [],
                'best_scores': [],
            }

    # ---- names ---------------------------------------------------------------------------------
    def metadata_name(self):
        return 'metadata.{}.yml'.format(self.name)

    def metadata_path(self):
        return os.path.join(self.folder, self.metadata_name())

    def ckpt_name(self, suffix):
        return '{}.{}.ckpt'.format(self.name, suffix)

    def ckpt_path(self, suffix):
        return os.path.join(self.folder, self.ckpt_name(suffix))

    # ---- metadata ------------------------------------------------------------------------------
    @staticmethod
    def _check_tracked_attrs(tracked_attrs):
        msg = 'tracked_attrs must be a list of attribute name strings or None'
        if isinstance(tracked_attrs, (list, tuple)):
            assert all(isinstance(a, str) for a in tracked_attrs), msg
        else:
            assert tracked_attrs is None, msg

    def _load_metadata(self):
        with open(self.metadata_path()) as fp:
            self.metadata = yaml.safe_load(fp)
        if str(self.metadata.get('version')) != CHECKPOINT_VERSION:
            raise ValueError('checkpoint version incompatible, please examine {} and make sure it is {}'
                             .format(self.metadata_path(), CHECKPOINT_VERSION))

    def _save_metadata(self):
        with open(self.metadata_path(), 'w') as fp:
            yaml.safe_dump(self.metadata, fp, default_flow_style=False)

    # ---- save ----------------------------------------------------------------------------------
    def _save_ckpt(self, suffix):
        attrs = self.metadata['tracked_attrs']
        assert attrs is not None, 'tracked_attrs must not be None for save(). ' \
                                  'Did you forget to restore from an existing checkpoint?'
        data = collections.OrderedDict()
        for attr in attrs:
            value = getattr(self.tracked_obj, attr)
            data[attr] = _to_host(value.state_dict()) if _stateful(value) else _to_host(value)
        with open(self.ckpt_path(suffix), 'wb') as fp:
            pickle.dump(data, fp)

    def save(self, score=None, global_steps=None, reload_metadata=False, **ckpt_info):
        if reload_metadata:
            self._load_metadata()
        meta = self.metadata
        meta['save_counter'] += 1
        if global_steps is None:
            global_steps = meta['save_counter']
        suffix = global_steps
        self._save_ckpt(suffix)
        meta['global_steps'] = global_steps
        meta['history_ckpt_files'] = [self.ckpt_name(suffix)] + \
            [f for f in meta['history_ckpt_files'] if f != self.ckpt_name(suffix)]
        for old in meta['history_ckpt_files'][meta['keep_history']:]:
            path = os.path.join(self.folder, old)
            if os.path.exists(path):
                os.remove(path)
        del meta['history_ckpt_files'][meta['keep_history']:]
        entry = {'score': score, 'global_steps': global_steps, 'save_counter': meta['save_counter'],
                 'time': time.time(), 'datetime': str(datetime.datetime.now())}
        entry.update(ckpt_info)
        meta['ckpt'][self.ckpt_name(suffix)] = entry
        if meta['keep_best'] > 0:
            assert score is not None, 'score cannot be None if keep_best is enabled'
            queue = _ScoreQueue(meta['keep_best'])
            to_delete = queue.set_queue(meta['best_scores'], meta['best_ckpt_files'])
            best_name = self.ckpt_name('best-{}'.format(suffix))
            evicted = queue.add(score, best_name)
            if evicted is None or evicted[1] != best_name:
                shutil.copy(self.ckpt_path(suffix), os.path.join(self.folder, best_name))
                meta['ckpt'][best_name] = dict(entry)
            if evicted:
                to_delete.append(evicted)
            for _, fname in to_delete:
                path = os.path.join(self.folder, fname)
                if os.path.exists(path):
                    os.remove(path)
                meta['ckpt'].pop(fname, None)
            meta['best_scores'], meta['best_ckpt_files'] = queue.get_scores_filepaths()
        self._save_metadata()

    # ---- restore -------------------------------------------------------------------------------
    def _restore(self, ckpt_file, check_ckpt_exists, folder):
        path = os.path.join(folder, ckpt_file)
        if not os.path.exists(path):
            if check_ckpt_exists:
                raise FileNotFoundError(path + ' missing.')
            return None
        with open(path, 'rb') as fp:
            data = pickle.load(fp)
        for attr in self.metadata['tracked_attrs']:
            value = getattr(self.tracked_obj, attr)
            if _stateful(value):
                value.load_state_dict(data[attr])
            else:
                setattr(self.tracked_obj, attr, data[attr])
        return path

    def restore(self, target, mode, reload_metadata=True, check_ckpt_exists=False,
                restore_folder=None):
        """target: int n = the n-th newest (or n-th best), or the global-steps suffix of a file"""
        assert mode in ('best', 'history')
        folder = os.path.expanduser(restore_folder) if restore_folder else self.folder
        if reload_metadata or restore_folder:
            keep = self.folder
            self.folder = folder
            try:
                self._load_metadata()
            finally:
                self.folder = keep
        meta = self.metadata
        if isinstance(target, int):
            assert target >= 0, 'target int should start from 0 for the last or best'
            files = meta['best_ckpt_files'] if mode == 'best' else meta['history_ckpt_files']
            if target < len(files):
                ckpt_file = files[target]
            elif check_ckpt_exists:
                raise FileNotFoundError('{} [{}] ckpt file missing'.format(mode.capitalize(), target))
            else:
                ckpt_file = '__DOES_NOT_EXIST__'
        else:
            assert '.ckpt' not in str(target), 'use restore_full_name() instead'
            ckpt_file = self.ckpt_name('best-{}'.format(target) if mode == 'best' else target)
        return self._restore(ckpt_file, check_ckpt_exists, folder)

    def restore_full_name(self, ckpt_file, check_ckpt_exists=True, restore_folder=None):
        folder = os.path.expanduser(restore_folder) if restore_folder else self.folder
        keep = self.folder
        self.folder = folder
        try:
            self._load_metadata()
        finally:
            self.folder = keep
        return self._restore(ckpt_file, check_ckpt_exists, folder)


class PeriodicCheckpoint(Checkpoint):
    """saves on every `period`-th call, at most once per `min_interval` seconds
    (checkpoint.py:316-354)"""

    def __init__(self, *args, period, min_interval=0, **kwargs):
        super().__init__(*args, **kwargs)
        assert period >= 1
        self.period = period
        self.min_interval = min_interval
        self._period_counter = 0
        self.last_update_time = time.time()

    def save(self, *args, **kwargs):
        self._period_counter += 1
        if self._period_counter % self.period == 0 and \
                time.time() - self.last_update_time >= self.min_interval:
            super().save(*args, **kwargs)
            self.last_update_time = time.time()
            return True
        return False

    def reset_period(self):
        self._period_counter = 0
