"""Host-side callers of the forward: chunker, feature-file dataset, the evaluation loop and its
multi-GPU sharding.  Counterparts of
  * `process_split` / `pad`                       /root/reference/data/tools.py:81-86,100-114
  * `UCF_Dataset` / `XD_Dataset` / `Shang_Dataset` /root/reference/data/dataset.py:8-127 (test mode)
  * `test()`                                      /root/reference/test.py:46-212,
                                                   train/ucf_test.py:16-216, train/xd_test.py:15-210
  * `compute_ano_auc`                             /root/reference/test.py:332-348
  * `run_test` (robustness sweep)                 /root/reference/test2.py:35-123 (`PerturbationSweep`)
The model is any callable with the reference's `model(img, ev, padding_mask, text, lengths)` contract.
"""
from __future__ import annotations

import collections
import concurrent.futures
import contextlib
import csv
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

# per-dataset class keys of the reference's class-wise tables (/root/reference/test.py:20-43)
CLASS_KEYS = {
    'ucfcrime': ['Abuse', 'Arrest', 'Arson', 'Assault', 'Burglary', 'Explosion', 'Fighting', 'RoadAccidents',
                 'Robbery', 'Shooting', 'Shoplifting', 'Stealing', 'Vandalism', 'Normal'],
    'xd': ['normal', 'fighting', 'shooting', 'riot', 'abuse', 'car accident', 'explosion'],
    'shang': ['car', 'chasing', 'fall', 'fighting', 'monocycle', 'robbery', 'running', 'skateboard',
              'throwing_object', 'vehicle', 'vaudeville', 'normal'],
    'msad': ['Normal', 'Assault', 'Explosion', 'Fighting', 'Fire', 'Object_falling', 'People_falling', 'Robbery',
             'Shooting', 'Traffic_accident', 'Vandalism', 'Water_incident'],
}
# event-feature path rule: str.replace over the WHOLE path (dataset.py:36,67,112)
EVENT_DIR = {'ucfcrime': 'event_thr_10', 'xd': 'event_thr_10', 'msad': 'event_thr_10', 'shang': 'event'}


def host_cpu_share() -> int:
    """CPUs this process may actually use: min(affinity, cgroup quota).  A GPU box can expose hundreds of
    hardware threads while a one-GPU job is capped at a small quota; a torch intra-op pool sized to
    os.cpu_count() then oversubscribes it and every small host-side tensor op crawls."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return int(os.environ.get("IEFVAD_CPU_THREADS", n))


# ------------------------------------------------------------------------------------------------
# chunker + dataset
# ------------------------------------------------------------------------------------------------
def process_split(feat: np.ndarray, length: int) -> Tuple[np.ndarray, int]:
    """Test-time chunking (tools.py:100-114): len < length -> one zero-padded [length, D] block;
    otherwise len//length + 1 blocks [n, length, D], the last zero padded -- an ALL-zero block when
    len % length == 0.  dtype is preserved."""
    n = int(feat.shape[0])
    if n < length:
        out = np.zeros((length, feat.shape[1]), dtype=feat.dtype)
        out[:n] = feat
        return out, n
    nchunk = n // length + 1
    out = np.zeros((nchunk * length, feat.shape[1]), dtype=feat.dtype)
    out[:n] = feat
    return out.reshape(nchunk, length, feat.shape[1]), n


class VideoFeatureDataset(torch.utils.data.Dataset):
    """Test-mode dataset over a `path,label` CSV of image-feature .npy files; the event file is found
    with the reference's replace rule.  Items match the reference's:
    (img [n,T,D] or [T,D], ev, label, length)."""

    def __init__(self, clip_dim: int, file_path: str, dataset: str = 'ucfcrime'):
        with open(file_path, newline='') as f:
            rows = list(csv.DictReader(f))
        self.paths = [r['path'] for r in rows]
        self.labels = [r['label'] for r in rows]
        self.clip_dim = clip_dim
        self.event_dir = EVENT_DIR[dataset]

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        p = self.paths[index]
        img = np.load(p)
        ev = np.load(p.replace('rgb', self.event_dir))
        img, n = process_split(img, self.clip_dim)
        ev, _ = process_split(ev, self.clip_dim)
        return torch.tensor(img), torch.tensor(ev), self.labels[index], n


def get_test_loader(args, dataset: Optional[str] = None):
    """`DataLoader(test_dataset, batch_size=1, shuffle=False)` (data/__getter__.py:27,39-40,66)."""
    ds = VideoFeatureDataset(args.visual_length, args.test_list, dataset or args.dataset)
    return torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)


# ------------------------------------------------------------------------------------------------
# metrics (sklearn, exactly the calls the reference makes)
# ------------------------------------------------------------------------------------------------
def compute_ano_auc(classwise_gt, classwise_roc, repeat_factor=16, normal_keys=('Normal',)):
    """test.py:332-348.  `normal_keys` covers the three variants of the filter: ('Normal',) in test.py:336,
    ('Normal', 'normal') in ucf_test.py:340, ('normal',) in xd_test.py:334."""
    from sklearn.metrics import roc_auc_score
    gt_abnormal, pred_abnormal = [], []
    for key in classwise_gt.keys():
        if key not in normal_keys and len(classwise_gt[key]) > 0:
            gt_abnormal.extend(np.concatenate(classwise_gt[key]).tolist())
            pred_abnormal.extend(np.concatenate(classwise_roc[key]).tolist())
    gt_abnormal = np.array(gt_abnormal)
    pred_abnormal = np.array(pred_abnormal)
    if len(np.unique(gt_abnormal)) > 1:
        return roc_auc_score(gt_abnormal, np.repeat(pred_abnormal, repeat_factor))
    return float('nan')


def evaluate_scores(scores: Sequence[np.ndarray], classes: Sequence[str], gt: np.ndarray, dataset: str,
                    verbose: bool = True, normal_keys=('Normal',), total_samples: bool = False,
                    log: Optional[Callable] = None) -> Dict[str, object]:
    """Metric tail of test() (test.py:155-175): ROC-AUC / AP on the x16-repeated snippet scores, Ano-AUC
    over the abnormal classes, per-class AUC/AP.  `scores` are per-video vectors in test-list order; gt
    is indexed by the running snippet offset (test.py:129,153)."""
    from sklearn.metrics import average_precision_score, roc_auc_score
    keys = CLASS_KEYS[dataset]
    cw_pred = {k: [] for k in keys}
    cw_gt = {k: [] for k in keys}
    st = 0
    for s, c in zip(scores, classes):
        cw_pred[c].append(np.asarray(s))
        cw_gt[c].append(gt[16 * st:16 * (st + len(s))])
        st += len(s)
    ap1 = np.concatenate([np.asarray(s) for s in scores]).tolist()
    roc = roc_auc_score(gt, np.repeat(ap1, 16))
    ap = average_precision_score(gt, np.repeat(ap1, 16))
    ano = compute_ano_auc(cw_gt, cw_pred, normal_keys=normal_keys)
    if verbose:
        print("AUC1: {:.2f}  AP1: {:.2f}".format(roc * 100, ap * 100))
        print("Ano-AUC: {:.2f}".format(ano * 100))
    if log is not None:      # ucf_test.py:158-162 / xd_test.py:155-159
        log({'test/AP1': ap, 'test/ROC1': roc, 'test/Ano-AUC': ano})
    per_class = {}
    for c in keys:
        cls_pred = np.concatenate(cw_pred[c])   # raises on an empty class exactly as test.py:166-167 does
        cls_gt = np.concatenate(cw_gt[c])
        if len(cls_gt) == 0 or sum(cls_gt) == 0:
            continue
        c_roc = roc_auc_score(cls_gt, np.repeat(cls_pred, 16))
        c_ap = average_precision_score(cls_gt, np.repeat(cls_pred, 16))
        if verbose and total_samples:      # ucf_test.py:173-174
            print(c, 'ROC: {:.2f}  AP: {:.2f}'.format(c_roc * 100, c_ap * 100), end='\t')
            print(f"Total Samples: {len(cls_gt)}")
        elif verbose:
            print(c, 'ROC: {:.2f}  AP: {:.2f}'.format(c_roc * 100, c_ap * 100))
        if log is not None:                # ucf_test.py:175-178 / xd_test.py:170-173
            log({'classwise/ROC/' + c: c_roc, 'classwise/AP/' + c: c_ap})
        per_class[c] = (c_roc, c_ap)
    if verbose:
        print('-------------------------------------------------')
    return {"roc": roc, "ap": ap, "ano_auc": ano, "per_class": per_class}


def device_auc_ap(scores: torch.Tensor, gt: torch.Tensor, repeat: int = 16) -> Tuple[float, float]:
    """ROC-AUC and average precision of `np.repeat(scores, repeat)` against frame-level `gt` -- the metric tail of
    test.py:158-159, which costs sklearn a sort of 16 x N points on the host -- by the library's `iefvad_auc_ap`
    (csrc/metrics.h: one radix sort of (score, positive frames of the snippet) pairs, a scan, a reduction over the tie
    groups; the repeat is never materialised).  Ties as sklearn: thresholds are the DISTINCT score values.
    `scores` [N] on a HIP device; `gt` [N * repeat] (any dtype, non-zero = anomalous; host or device)."""
    from . import lib as _lib
    import ctypes as C
    if not scores.is_cuda:
        raise RuntimeError("device_auc_ap runs on a HIP device only (iefvad_auc_ap); on the host use evaluate_scores (sklearn)")
    s = scores.reshape(-1).to(torch.float32).contiguous()
    g = torch.as_tensor(gt).reshape(-1)
    if g.numel() != s.numel() * repeat:
        raise ValueError("gt must hold `repeat` frames per snippet")
    # the kernel counts every non-zero byte as an anomalous frame: one-byte ground truth (uint8 / bool) goes in as it is
    g = g if g.dtype in (torch.uint8, torch.bool) else (g != 0).to(torch.uint8)
    g = g.to(s.device, non_blocking=True).contiguous()
    lib = _lib.load_library()
    n = s.numel()
    with torch.cuda.device(s.device):
        ws = torch.empty(lib.iefvad_auc_ap_workspace_bytes(n) + 256, dtype=torch.uint8, device=s.device)
        off = (-ws.data_ptr()) % 256
        out = torch.empty(2, dtype=torch.float64, device=s.device)
        rc = lib.iefvad_auc_ap(C.c_void_p(s.data_ptr()), C.c_void_p(g.data_ptr()), n, repeat, C.c_void_p(out.data_ptr()),
                               C.c_void_p(out.data_ptr() + 8), C.c_void_p(ws.data_ptr() + off), ws.numel() - off,
                               C.c_void_p(torch.cuda.current_stream(s.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("iefvad_auc_ap: " + _lib.last_error())
    auc, ap = out.cpu().tolist()
    return auc, ap


# ------------------------------------------------------------------------------------------------
# the evaluation loop
# ------------------------------------------------------------------------------------------------
def _has_nan(t: torch.Tensor) -> bool:
    """`torch.isnan(t).any()` (test.py:90,93) without a full boolean pass in the common case: a NaN element makes the sum
    NaN, so only a NaN sum (a real NaN, or +inf and -inf meeting) pays for the exact scan.  On the host this check was a
    third of the per-video loop's time."""
    if not t.is_floating_point():
        return False
    s = t.sum(dtype=torch.float32) if t.dtype in (torch.float16, torch.bfloat16) else t.sum()
    return bool(torch.isnan(s)) and bool(torch.isnan(t).any())


def _unpack_item(item, maxlen, dataset, label_map):
    """Shape rule of test.py:77-88 applied to one DataLoader item (batch_size=1)."""
    img = item[0].squeeze(0)
    ev = item[1].squeeze(0)
    cls = item[2][0] if isinstance(item[2], (list, tuple)) else item[2]
    if label_map is not None and (dataset == 'xd' or isinstance(label_map, _AlwaysRemap)):
        cls = label_map[cls.split('-')[0]]            # xd_test.py:68 / test.py:80-81
    length = int(item[3])
    if length < maxlen:
        img = img.unsqueeze(0)
        ev = ev.unsqueeze(0)
    if _has_nan(img):                                 # conditional nan_to_num, test.py:90-95
        img = torch.nan_to_num(img, nan=0.0)
    if _has_nan(ev):
        ev = torch.nan_to_num(ev, nan=0.0)
    return img, ev, cls, length


def _unpack_rows(item, maxlen, dataset, label_map):
    """One DataLoader item (batch_size=1, chunked and zero padded by the loader as the reference's is) reduced to what
    crosses the boundary in the ragged path: per modality the item's own [..., D] tensor, whose first `length` rows are the
    valid ones (no view is built: two tensor ops per video were a third of the loop's host time on short videos)."""
    cls = item[2][0] if isinstance(item[2], (list, tuple)) else item[2]
    if label_map is not None and (dataset == 'xd' or isinstance(label_map, _AlwaysRemap)):
        cls = label_map[cls.split('-')[0]]            # xd_test.py:68 / test.py:80-81
    length = int(item[3])
    img, ev = item[0], item[1]
    if not img.is_contiguous():
        img = img.contiguous()
    if not ev.is_contiguous():
        ev = ev.contiguous()
    return img, ev, cls, length


class _PinnedStager:
    """Host -> device hand-over of the evaluation loop.  The reference does `img.to(device)` on pageable tensors
    (test.py:90-95), a synchronous staged copy at a few GB/s that blocks the host for longer than a one-chunk forward
    takes on the device.  Here the chunks of a forward are gathered into one of two reusable PINNED buffers (one host
    memcpy, which also does the torch.cat of a packed batch) and sent with an asynchronous copy on the current stream;
    a buffer is reused only after the event behind its last copy has completed."""

    def __init__(self, device, slots: int = 2, threads: int = 0):
        self.device = torch.device(device)
        self.threads = threads or max(1, min(8, host_cpu_share() // 2))      # host threads of one staging copy
        self.bufs = [[None, None] for _ in range(slots)]      # per slot: image / event pinned byte buffers
        self.events = [None] * slots
        self.turn = 0

    def _buffer(self, slot, m, nbytes):
        b = self.bufs[slot][m]
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, pin_memory=True)
            self.bufs[slot][m] = b
        return b

    def upload(self, imgs, evs, dt):
        slot = self.turn
        self.turn = (self.turn + 1) % len(self.bufs)
        if self.events[slot] is not None:
            self.events[slot].synchronize()
        out = []
        for m, parts in enumerate((imgs, evs)):
            n = sum(int(p.shape[0]) for p in parts)
            shape = (n,) + tuple(parts[0].shape[1:])
            count = n * int(parts[0].shape[1]) * int(parts[0].shape[2])
            esize = torch.empty(0, dtype=dt).element_size()
            host = self._buffer(slot, m, count * esize)[:count * esize].view(dt).view(shape)
            off = 0
            for p in parts:
                host[off:off + p.shape[0]].copy_(p)          # casts when the batch was widened
                off += p.shape[0]
            out.append(host.to(self.device, non_blocking=True))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.events[slot] = ev
        return out[0], out[1]


class _RowStager:
    """Host -> device hand-over of VALID rows (the packed evaluation loop through `MMFMIL.forward_videos`): the feature rows
    of a batch of videos are copied, without their zero padding, into one of `slots` reusable pinned buffers -- by a few
    threads when the batch is large -- and sent with one asynchronous copy per modality on the current stream.  Nothing on
    the host reads the rows: the NaN scan of test.py:90-95 happens on the device (csrc/ragged.h)."""

    def __init__(self, device, slots: int = 2, threads: int = 0):
        self.device = torch.device(device)
        self.threads = threads or max(1, min(8, host_cpu_share() // 2))      # host threads of one staging copy
        self.bufs = [[None, None] for _ in range(slots)]
        self.events = [None] * slots
        self.turn = 0

    def _buffer(self, slot, m, nbytes):
        b = self.bufs[slot][m]
        if b is None or b.numel() < nbytes:
            # pinning memory costs milliseconds per 100 MB: grow EVERY slot now (the first pass over a list is the warm-up,
            # later passes must not meet a slot that was never used at this size), with head room for uneven batches
            size = max(int(nbytes * 1.5), 1 << 22)
            for s_ in range(len(self.bufs)):
                for m_ in range(2):
                    old = self.bufs[s_][m_]
                    if old is None or old.numel() < size:
                        if self.events[s_] is not None:
                            self.events[s_].synchronize()
                        self.bufs[s_][m_] = torch.empty(size, dtype=torch.uint8, pin_memory=True)
            b = self.bufs[slot][m]
        return b

    def stage(self, imgs, evs, dt, lens=None):
        """Host half of `upload`: the rows of all videos into the next pinned slot.  Touches no stream (it only waits for the
        slot's previous copy), so a worker thread can run it while the caller's thread enqueues the previous batch.
        `lens` given: imgs / evs are contiguous [..., D] tensors whose FIRST lens[i] rows are taken."""
        slot = self.turn
        self.turn = (self.turn + 1) % len(self.bufs)
        if self.events[slot] is not None:
            self.events[slot].synchronize()
        D = int(imgs[0].shape[-1])
        if lens is None:
            lens = [int(p.shape[0]) for p in imgs]
        total = sum(lens)
        esize = torch.empty(0, dtype=dt).element_size()
        hosts = [self._buffer(slot, m, total * D * esize)[:total * D * esize].view(dt).view(total, D) for m in range(2)]
        if all(p.dtype == dt and p.is_contiguous() for parts in (imgs, evs) for p in parts):
            # one library call per modality: the rows of all videos gathered into the pinned buffer by a few host threads
            # (no GIL, no per-video tensor op: ~90 us of torch dispatch per video was 10x the forward's time on short videos)
            import ctypes as C
            from . import lib as _lib
            lib = _lib.load_library()
            n = len(lens)
            sizes = (C.c_size_t * n)(*[l * D * esize for l in lens])
            for host, parts in zip(hosts, (imgs, evs)):
                ptrs = (C.c_void_p * n)(*[p.data_ptr() for p in parts])
                if lib.iefvad_host_gather(C.c_void_p(host.data_ptr()), ptrs, sizes, n, self.threads) != 0:
                    raise RuntimeError("iefvad_host_gather: " + _lib.last_error())
        else:                                                     # a widened (mixed-dtype) batch: torch casts while copying
            off = 0
            for i, l in enumerate(lens):
                hosts[0][off:off + l].copy_(imgs[i].reshape(-1, D)[:l])
                hosts[1][off:off + l].copy_(evs[i].reshape(-1, D)[:l])
                off += l
        return slot, hosts

    def send(self, staged):
        """Device half: one asynchronous copy per modality on the current stream.  Returns two [sum(len), D] device tensors."""
        slot, hosts = staged
        out = [h.to(self.device, non_blocking=True) for h in hosts]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.events[slot] = ev
        return out[0], out[1]

    def upload(self, imgs, evs, dt, lens=None):
        """imgs / evs: per video [len, D] host tensors (views are fine), or with `lens` contiguous [..., D] tensors of which the
        first lens[i] rows count.  Returns two [sum(len), D] device tensors."""
        return self.send(self.stage(imgs, evs, dt, lens))


def score_loader(model: Callable, test_loader: Iterable, maxlen: int, device, dataset: str = 'ucfcrime',
                 label_map=None, batch_chunks: int = 0, skip_empty_chunks: bool = True, lanes: int = 1,
                 ragged: Optional[bool] = None, host_list: bool = True, host_list_bytes: int = 1 << 30, wire_bf16: bool = False):
    """Per-video sigmoid scores and mean fusion weights, in loader order.

    batch_chunks == 0: one forward per video with B = that video's chunk count -- the reference's call
    pattern (test.py:76-117).  batch_chunks > 0: chunks of consecutive videos are packed into one
    forward of up to `batch_chunks` chunks; legal because chunks are independent batch rows
    (imf_vad.py:115 attends within a chunk), and the trailing all-zero chunk of a len % 256 == 0 video,
    whose rows the reference slices away (test.py:121), is not computed when `skip_empty_chunks`.

    `ragged` (default: whenever the model has `forward_videos`, i.e. `iefvad_amd.MMFMIL` on a HIP device, and batch_chunks > 0):
    the packed batches go through `MMFMIL.forward_videos` -- only the VALID rows of every video are staged and uploaded, the
    chunker, the conditional nan_to_num (test.py:90-95) and the `[0:len]` slicing run on the device, and no stage computes the
    padding (one pad row per chunk stands for all of them, csrc/ragged.h).  Same scores (bit for bit in the f32 and bf16 modes).

    `host_list` (default; ragged mode with host-resident loader tensors): the walk over the list happens inside the library
    (`iefvad_forward_videos_host`): Python hands over each video's tensor and length, the library packs, stages, sends and scores
    pass by pass with its own worker thread and copy stream -- one call per `host_list_bytes` of features.  `lanes` is not used then.
    `wire_bf16` (list path, fp32 features, a compute="bf16" model): the rows are rounded to bf16 while they are staged and half the
    bytes cross PCIe (`MMFMIL.forward_videos_host(wire_dtype=torch.bfloat16)`); off by default.

    lanes > 1 (HIP devices, `iefvad_amd.MMFMIL`): consecutive forwards go round-robin to `lanes` HIP streams, each with a
    lane of the model (`MMFMIL.lanes`: same parameters, own library handle and workspace) and its own pinned staging.
    Same kernels on the same inputs, so every score is bit-identical to lanes = 1; what changes is that the one- and
    two-chunk forwards of the per-video pattern, each of which fills a fraction of the chip, overlap."""
    classes: List[str] = []
    pend: List[Tuple[torch.Tensor, torch.Tensor, int]] = []
    pend_chunks = 0
    # results stay on the model's device until the loop is over: the reference synchronises three times per video
    # (`.cpu()` of prob / w_i / w_e, test.py:119-151); here the forwards are enqueued back to back and the scores of
    # all videos come back in ONE device-to-host copy at the end
    dev_prob: List[torch.Tensor] = []
    dev_wi: List[torch.Tensor] = []
    dev_we: List[torch.Tensor] = []
    spans: List[Tuple[int, int]] = []          # (offset into the concatenated device vectors, valid length) per video
    total = 0

    on_gpu = torch.device(device).type == 'cuda'
    nl = lanes if (on_gpu and lanes > 1 and hasattr(model, 'lanes')) else 1
    models = model.lanes(nl) if nl > 1 else [model]
    if ragged is None:
        ragged = on_gpu and batch_chunks > 0 and skip_empty_chunks and hasattr(model, 'forward_videos')
    elif ragged and not (on_gpu and batch_chunks > 0 and hasattr(model, 'forward_videos')):
        raise ValueError("ragged=True needs a HIP device, batch_chunks > 0 and a model with forward_videos")
    # staging buffers are pinned allocations: they live with the lane (model object) across calls, not with the call
    stagers = []
    for m in models:
        if not on_gpu:
            stagers.append(None)
            continue
        cache = m.__dict__.setdefault("_stagers", {}) if hasattr(m, "__dict__") else {}
        key = ("rows" if ragged else "chunks", str(device))
        if key not in cache:
            cache[key] = _RowStager(device) if ragged else _PinnedStager(device)
        stagers.append(cache[key])
    streams = [torch.cuda.Stream(device=device) for _ in range(nl)] if nl > 1 else [None]
    if nl > 1:
        for s in streams:
            s.wait_stream(torch.cuda.current_stream(device))     # e.g. a `model.to(device)` still in flight
    nflush = 0
    # packed valid-row loop: the staging copy of batch k + 1 (host threads inside the library, no GIL) runs on a worker
    # thread while this thread sends batch k and enqueues its forward; batches complete in order
    inflight: collections.deque = collections.deque()
    stage_pool = concurrent.futures.ThreadPoolExecutor(max_workers=1) if ragged else None

    def run_oldest():
        nonlocal total
        fut, k, lens = inflight.popleft()
        staged = fut.result()
        ctx = torch.cuda.stream(streams[k]) if nl > 1 else contextlib.nullcontext()
        with ctx:
            img, ev = stagers[k].send(staged)
            out = models[k].forward_videos(img, ev, lens, nan_to_num=True)
        dev_prob.append(out['logits'])
        dev_wi.append(out['w_i_mean'])
        dev_we.append(out['w_e_mean'])
        off = total
        for n in lens:
            spans.append((off, n))
            off += n
        total = off

    def flush():
        nonlocal nflush, pend, pend_chunks
        if not pend:
            return
        k = nflush % nl
        nflush += 1
        if ragged:
            dts = {p[0].dtype for p in pend} | {p[1].dtype for p in pend}
            dt = torch.float32 if len(dts) > 1 else pend[0][0].dtype
            if len(dts) > 1:      # a narrower video is widened at staging: its own dtype's inf -> max rule applies first (test.py:90-95)
                pend = [tuple(torch.nan_to_num(t, nan=0.0) if (t.dtype != dt and _has_nan(t)) else t for t in (p[0], p[1])) + (p[2],)
                        for p in pend]
            lens = [n for _, _, n in pend]
            fut = stage_pool.submit(stagers[k].stage, [p[0] for p in pend], [p[1] for p in pend], dt, lens)
            inflight.append((fut, k, lens))
            pend, pend_chunks = [], 0
            while len(inflight) > 1:
                run_oldest()
            return
        if nl > 1:
            with torch.cuda.stream(streams[k]):
                flush_on(models[k], stagers[k])
        else:
            flush_on(models[0], stagers[0])

    def flush_on(model, stager):
        nonlocal pend, pend_chunks, total
        # dtype as the files hold it (dataset.py:37-38,49-50); the model widens with `.to(torch.float)` per modality
        # (imf_vad.py:41-42), so a batch that mixes dtypes -- across videos OR between the two modalities -- is widened
        # to fp32 here rather than narrowed to the first tensor's type
        dts = {p[0].dtype for p in pend} | {p[1].dtype for p in pend}
        dt = torch.float32 if len(dts) > 1 else pend[0][0].dtype
        if stager is not None:
            img, ev = stager.upload([p[0] for p in pend], [p[1] for p in pend], dt)
        else:
            img = torch.cat([p[0].to(dt) for p in pend], dim=0).to(device)
            ev = torch.cat([p[1].to(dt) for p in pend], dim=0).to(device)
        out = model(img, ev, None, None, None)
        logits = out['logits'].reshape(-1)
        if 'w_i_mean' in out:
            wi, we = out['w_i_mean'].reshape(-1), out['w_e_mean'].reshape(-1)
        else:
            wi = out['w_i'].reshape(-1, out['w_i'].shape[-1]).mean(dim=-1)     # test.py:131-136
            we = out['w_e'].reshape(-1, out['w_e'].shape[-1]).mean(dim=-1)
        dev_prob.append(logits.float())                      # sigmoid once, on the concatenated logits, after the loop
        dev_wi.append(wi.float())
        dev_we.append(we.float())
        off = total
        for ci, _, n in pend:
            spans.append((off, n))                           # logits1[0:len_cur] -> sigmoid, test.py:119-121
            off += ci.shape[0] * maxlen
        total = off
        pend, pend_chunks = [], 0

    # packed valid-row loop with the walk inside the library (`MMFMIL.forward_videos_host`, csrc/hostpipe.h): this thread only
    # collects each video's host tensor and length; one library call per `host_list_bytes` of features stages, sends and scores
    # them pass by pass.  Same batches, same kernels as the loop below (which remains for models without the entry, lanes > 1
    # callers that ask for it with host_list=False, and device-resident loaders).
    use_list = bool(ragged and host_list and hasattr(model, 'forward_videos_host'))
    if use_list:
        group: List[Tuple[torch.Tensor, torch.Tensor, int]] = []
        gbytes = 0

        def flush_list():
            nonlocal group, gbytes, total
            if not group:
                return
            wire = torch.bfloat16 if (wire_bf16 and group[0][0].dtype == torch.float32) else None
            out = model.forward_videos_host([g[0] for g in group], [g[1] for g in group], [g[2] for g in group], nan_to_num=True,
                                            batch_chunks=batch_chunks, wire_dtype=wire)
            dev_prob.append(out['logits'])
            dev_wi.append(out['w_i_mean'])
            dev_we.append(out['w_e_mean'])
            off = total
            for _, _, n in group:
                spans.append((off, n))
                off += n
            total = off
            group, gbytes = [], 0

        with torch.no_grad():
            for item in test_loader:
                img, ev, cls, n = _unpack_rows(item, maxlen, dataset, label_map)
                classes.append(cls)
                if img.is_cuda or ev.is_cuda:
                    raise ValueError("host_list=True expects the loader's tensors in host memory")
                if img.dtype != ev.dtype or img.dtype not in (torch.float32, torch.float16, torch.bfloat16):
                    # the model widens both with `.to(torch.float)` (imf_vad.py:41-42); the NaN rule applies in the file's own dtype first
                    img, ev = (torch.nan_to_num(t, nan=0.0) if _has_nan(t) else t for t in (img, ev))
                    img, ev = img.float(), ev.float()
                if group and group[0][0].dtype != img.dtype:
                    flush_list()                                  # a library call takes one feature dtype
                group.append((img, ev, n))
                gbytes += 2 * n * img.shape[-1] * img.element_size()
                if gbytes >= host_list_bytes:
                    flush_list()
            flush_list()
        test_loader = ()
        ragged = False

    with torch.no_grad():
        try:
            for item in test_loader:
                if ragged:
                    img, ev, cls, n = _unpack_rows(item, maxlen, dataset, label_map)
                    classes.append(cls)
                    pend.append((img, ev, n))
                    pend_chunks += n // maxlen + (1 if n % maxlen else 0) if n >= maxlen else 1
                    if pend_chunks >= batch_chunks:
                        flush()
                    continue
                img, ev, cls, n = _unpack_item(item, maxlen, dataset, label_map)
                classes.append(cls)
                if batch_chunks > 0 and skip_empty_chunks and n >= maxlen and n % maxlen == 0:
                    img, ev = img[:-1], ev[:-1]               # the all-zero chunk (tools.py:105-112)
                pend.append((img, ev, n))
                pend_chunks += img.shape[0]
                if batch_chunks <= 0 or pend_chunks >= batch_chunks:
                    flush()
            flush()
            while inflight:
                run_oldest()
        finally:
            if stage_pool is not None:         # also when a forward raised: the worker thread must not outlive the call
                stage_pool.shutdown(wait=True)
        if nl > 1:
            for s in streams:
                torch.cuda.current_stream(device).wait_stream(s)
        if dev_prob:
            # ONE device-to-host copy for the three vectors (sigmoid on the device: logits1[0:len_cur] -> sigmoid, test.py:119-121)
            allv = torch.stack([torch.sigmoid(torch.cat(dev_prob)), torch.cat(dev_wi), torch.cat(dev_we)]).cpu().numpy()
            prob, wi, we = allv[0], allv[1], allv[2]
        else:
            prob = wi = we = np.zeros(0, np.float32)
    # per-video views of the three host vectors (contiguous spans in loader order; padded routes leave gaps between them)
    scores = [prob[o:o + n] for o, n in spans]
    wi_means = [wi[o:o + n] for o, n in spans]
    we_means = [we[o:o + n] for o, n in spans]
    return scores, classes, wi_means, we_means


class FeatureFilePipeline:
    """Streaming front end for real feature files.  The `.npy` headers are parsed first (shape
    and dtype without touching the payload), the chunks of consecutive videos are laid out into batches of
    >= `batch_chunks` chunks, and worker threads then read() each video's image / event payload STRAIGHT into its
    slot of a pinned host staging buffer (one copy, GIL released, in parallel); a finished batch goes to the
    device on a side stream, so file I/O, PCIe and the forward of the previous batch overlap.  Order is preserved.
    Semantics match the reference loader + test() prologue: dtype preserved from disk (dataset.py:37-38,49-50;
    a batch mixing dtypes is widened to fp32, which `.to(torch.float)` does anyway), conditional nan_to_num
    (test.py:90-95), and the all-zero trailing chunk of a len % 256 == 0 video is dropped (its rows are sliced
    away by test.py:121)."""

    def __init__(self, paths: Sequence[str], labels: Sequence[str], clip_dim: int, event_dir: str, device,
                 batch_chunks: int = 64, workers: int = 2, prefetch: int = 2, ragged: bool = False):
        self.paths, self.labels = list(paths), list(labels)
        self.clip_dim, self.event_dir, self.device = clip_dim, event_dir, torch.device(device)
        self.batch_chunks, self.workers, self.prefetch = batch_chunks, workers, max(1, prefetch)
        # ragged: batches are the videos' VALID rows only, [sum(len), D] per modality, for `MMFMIL.forward_videos`; no zero
        # padding is written and nothing on the host scans the payload (the NaN rule of test.py:90-95 runs on the device)
        self.ragged = ragged

    @staticmethod
    def _header(path):
        """(shape, dtype, payload offset) of a .npy file without touching the payload."""
        with open(path, 'rb') as f:
            ver = np.lib.format.read_magic(f)
            rd = np.lib.format.read_array_header_1_0 if ver == (1, 0) else np.lib.format.read_array_header_2_0
            shape, fortran, dtype = rd(f)
            if fortran or len(shape) != 2:
                raise ValueError(f"{path}: expected a C-ordered [len, D] array")
            return shape, dtype, f.tell()

    def _open(self, idx):
        p = self.paths[idx]
        pe = p.replace('rgb', self.event_dir)
        hi, he = self._header(p), self._header(pe)
        if tuple(hi[0]) != tuple(he[0]):
            # the reference would fail later, at the model's residual add of mismatched lengths; fail at the file
            raise ValueError(f"{pe}: event features {tuple(he[0])} do not match the image features {tuple(hi[0])} of {p}")
        return (p,) + hi, (pe,) + he

    @staticmethod
    def _fill(dst: np.ndarray, src, scan: bool = True):
        """One read() of the payload from the page cache straight into the pinned staging rows (no intermediate
        array, no page faults of a memory map); a dtype mismatch (mixed-dtype batch) goes through a temporary."""
        path, shape, dtype, offset = src
        with open(path, 'rb', buffering=0) as f:
            f.seek(offset)
            if dtype == dst.dtype:
                got = f.readinto(memoryview(dst.reshape(-1).view(np.uint8)))
                if got != dst.nbytes:
                    raise IOError(f"{path}: short read ({got} of {dst.nbytes} bytes)")
            else:
                tmp = np.fromfile(f, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
                np.copyto(dst, tmp, casting='unsafe')
        if not scan and dtype == dst.dtype:
            return            # ragged path, payload in its own dtype: the device decides (csrc/ragged.h)
        if np.issubdtype(dst.dtype, np.floating) and np.isnan(np.sum(dst, dtype=np.float32)) and np.isnan(dst).any():
            # conditional nan_to_num (test.py:90-95): NaN -> 0, +-inf -> the SOURCE dtype's max / min
            fi = np.finfo(dtype)
            np.nan_to_num(dst, copy=False, nan=0.0, posinf=float(fi.max), neginf=float(fi.min))

    @classmethod
    def _fill_many(cls, jobs, scan: bool = True):
        for dst, src in jobs:
            cls._fill(dst, src, scan)

    def batches(self):
        """Yields (img [B,T,D] device tensor, ev, [(video index, n snippets, n chunks), ...])."""
        from concurrent.futures import ThreadPoolExecutor
        T = self.clip_dim
        copy_stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        pin = self.device.type == 'cuda'
        with ThreadPoolExecutor(self.workers) as pool:
            maps = list(pool.map(self._open, range(len(self.paths))))      # headers only
            plans, cur, cur_chunks = [], [], 0
            for idx, (mi, _) in enumerate(maps):
                n = int(mi[1][0])
                nch = (n // T + (1 if n % T else 0)) if n >= T else 1
                cur.append((idx, n, nch))
                cur_chunks += nch
                if cur_chunks >= self.batch_chunks:
                    plans.append(cur)
                    cur, cur_chunks = [], 0
            if cur:
                plans.append(cur)

            def start(plan):
                dts = {maps[i][0][2] for i, _, _ in plan} | {maps[i][1][2] for i, _, _ in plan}
                dt = np.dtype(np.float32) if len(dts) > 1 else next(iter(dts))
                D = int(maps[plan[0][0]][0][1][1])
                nchunks = sum(nch for _, _, nch in plan)
                tdt = torch.from_numpy(np.zeros(0, dt)).dtype
                if self.ragged:
                    nrows = sum(n for _, n, _ in plan)
                    hi = torch.empty(nrows, D, dtype=tdt, pin_memory=pin)
                    he = torch.empty(nrows, D, dtype=tdt, pin_memory=pin)
                    ni, ne = hi.numpy(), he.numpy()
                    jobs, off = [], 0
                    for idx, n, nch in plan:
                        jobs.append((ni[off:off + n], maps[idx][0]))
                        jobs.append((ne[off:off + n], maps[idx][1]))
                        off += n
                    nt = max(1, min(self.workers, len(jobs)))
                    return hi, he, [pool.submit(self._fill_many, jobs[w::nt], False) for w in range(nt)], plan
                hi = torch.empty(nchunks, T, D, dtype=tdt, pin_memory=pin)
                he = torch.empty(nchunks, T, D, dtype=tdt, pin_memory=pin)
                ni, ne = hi.numpy().reshape(-1, D), he.numpy().reshape(-1, D)
                futs, off = [], 0
                jobs = []
                for idx, n, nch in plan:
                    jobs.append((ni[off * T:off * T + n], maps[idx][0]))
                    jobs.append((ne[off * T:off * T + n], maps[idx][1]))
                    ni[off * T + n:(off + nch) * T] = 0          # zero padding of the video's last chunk only
                    ne[off * T + n:(off + nch) * T] = 0
                    off += nch
                # a few coarse tasks per batch: a page-cache read() runs at ~12 GB/s on one thread, so the Python
                # per-task overhead (and the GIL) of hundreds of tiny tasks would dominate
                nt = max(1, min(self.workers, len(jobs)))
                for w in range(nt):
                    futs.append(pool.submit(self._fill_many, jobs[w::nt]))
                return hi, he, futs, plan

            inflight = [start(p) for p in plans[:self.prefetch]]
            nxt = len(inflight)
            while inflight:
                hi, he, futs, plan = inflight.pop(0)
                for f in futs:
                    f.result()
                if nxt < len(plans):
                    inflight.append(start(plans[nxt]))
                    nxt += 1
                if copy_stream is not None:
                    with torch.cuda.stream(copy_stream):
                        di = hi.to(self.device, non_blocking=True)
                        de = he.to(self.device, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(copy_stream)
                    torch.cuda.current_stream(self.device).wait_event(done)
                    di.record_stream(torch.cuda.current_stream(self.device))
                    de.record_stream(torch.cuda.current_stream(self.device))
                else:
                    di, de = hi, he
                yield di, de, plan


def evaluate_files(args, model, gt, device, dataset: Optional[str] = None, batch_chunks: int = 64, workers: int = 2,
                   device_metrics: bool = True, verbose: bool = False):
    """End-to-end evaluation from a `path,label` CSV of feature files: streaming loader -> batched forward ->
    ordered scores -> AUC / AP (on the device when `device_metrics`).  Returns a dict with the metrics, the
    per-video scores and a wall-clock phase breakdown."""
    import time
    dataset = dataset or args.dataset
    with open(args.test_list, newline='') as f:
        rows = list(csv.DictReader(f))
    paths, labels = [r['path'] for r in rows], [r['label'] for r in rows]
    ragged = torch.device(device).type == 'cuda' and hasattr(model, 'forward_videos')
    pipe = FeatureFilePipeline(paths, labels, args.visual_length, EVENT_DIR[dataset], device, batch_chunks, workers,
                               ragged=ragged)
    model.eval()
    t0 = time.perf_counter()
    outs, metas = [], []
    with torch.no_grad():
        for img, ev, meta in pipe.batches():
            if ragged:       # valid rows only; chunker, NaN rule and [0:len] slicing on the device (csrc/ragged.h)
                outs.append(model.forward_videos(img, ev, [n for _, n, _ in meta])['logits'])
            else:
                o = model(img, ev, None, None, None)
                outs.append(o['logits'].reshape(img.shape[0], -1))
            metas.append(meta)
    T = args.visual_length
    pieces = []
    for lg, meta in zip(outs, metas):
        if ragged:
            pieces.append(lg)
            continue
        off = 0
        for _, n, nch in meta:
            pieces.append(lg[off:off + nch].reshape(-1)[:n])
            off += nch
    scores_dev = torch.sigmoid(torch.cat(pieces))
    if scores_dev.is_cuda:
        torch.cuda.synchronize(scores_dev.device)
    t1 = time.perf_counter()
    res: Dict[str, object] = {}
    device_metrics = device_metrics and scores_dev.is_cuda      # the metric kernels are the library's; a CPU run uses sklearn below
    if device_metrics:
        roc, ap = device_auc_ap(scores_dev, torch.as_tensor(gt))
        res.update(roc=roc, ap=ap)
    scores_host = scores_dev.float().cpu().numpy()
    t2 = time.perf_counter()
    lens = [n for meta in metas for _, n, _ in meta]
    offs = np.concatenate([[0], np.cumsum(lens)])
    per_video = [scores_host[offs[i]:offs[i + 1]] for i in range(len(lens))]
    if not device_metrics:
        res.update(evaluate_scores(per_video, labels, gt, dataset, verbose=verbose))
    res.update(scores=per_video, classes=labels, snippets=int(offs[-1]),
               seconds={"load+h2d+forward": t1 - t0, "metrics": t2 - t1})
    return res


def _run_test(args, model, test_loader, maxlen, gt, device, label_map, attn, vis, normal_keys, total_samples, log,
              batch_chunks, lanes):
    """The body the three `test()` functions of the reference share (test.py:57-212): model.to / eval, the per-video
    loop, the metric tail, the prints.  What differs between the files is passed in: the Ano-AUC filter, the
    "Total Samples" suffix of ucf_test.py:173-174 and the `wandb.log` calls (ucf_test.py:158-162,175-178 /
    xd_test.py:155-159,170-173), which go to `log` when the caller supplies one (wandb itself is out of scope)."""
    model.to(device)
    model.eval()
    if batch_chunks is None:
        # default of the three entries: the packed route (the list walked inside the library, valid rows only) wherever it gives the
        # SAME BITS as one forward per video -- compute "f32" (every tiling sums k in one order) and "bf16" (ring and row-block kernels
        # are bit-identical); "bf16x6" / "fp16x3" pick their kernels by batch size, so they keep the reference's per-video pattern unless
        # the caller asks (batch_chunks=64: scores move at the 1e-7 level, inside the fp32 gates).  Five times the per-video rate on a
        # UCF-sized list.
        packed_same_bits = (getattr(model, "compute", None) in ("f32", "bf16") and torch.device(device).type == "cuda" and lanes == 1)
        batch_chunks = 64 if packed_same_bits else 0
    scores, classes, wi, we = score_loader(model, test_loader, maxlen, device, args.dataset, label_map, batch_chunks,
                                           lanes=lanes)
    res = evaluate_scores(scores, classes, gt, args.dataset, verbose=True, normal_keys=normal_keys,
                          total_samples=total_samples, log=log)
    last = dict(res, scores=scores, classes=classes, w_i_mean=wi, w_e_mean=we)
    if vis:
        print("[iefvad_amd] vis=True: plotting (test.py:177-207) is outside the hot-path scope; skipped")
    if attn:     # the reference returns the empty list it never fills (test.py:73,209-210)
        return (res["roc"], res["ap"], [], classes), last
    return (res["roc"], res["ap"]), last


def test(args, model, test_loader, maxlen, prompt_text, gt, device, attn=False, vis=False, label_map=None,
         batch_chunks: Optional[int] = None, normal_keys=('Normal',), lanes: int = 1):
    """Counterpart of ROOT `test.py`'s `test()` (test.py:46-56; call site test.py:380-390): same positional order
    (..., device, attn, vis), Ano-AUC over every class but 'Normal' (test.py:336).  `label_map` is keyword-only in
    spirit: root test.py reads a global for the xd remap (test.py:81).  Returns (ROC1, AP1), or
    (ROC1, AP1, attn_weights, labels) when attn=True.  For `train/ucf_test.py` and `train/xd_test.py` -- whose
    positional orders differ from this one and from each other -- use `ucf_test` / `xd_test` below."""
    ret, test.last_result = _run_test(args, model, test_loader, maxlen, gt, device, label_map, attn, vis, normal_keys,
                                      False, None, batch_chunks, lanes)
    return ret


def ucf_test(args, model, test_loader, maxlen, prompt_text, gt, device, attn=False, vis=False, *, log=None,
             batch_chunks: Optional[int] = None, lanes: int = 1):
    """Drop-in for `train/ucf_test.py`'s `test` (ucf_test.py:16-26; call site ucf_train.py:130-139): positional order
    (..., device, attn, vis); Ano-AUC excludes BOTH 'Normal' and 'normal' (ucf_test.py:340); the per-class lines carry
    "Total Samples" (ucf_test.py:173-174).  `log` (e.g. `wandb.log`) receives the dicts the reference logs."""
    ret, ucf_test.last_result = _run_test(args, model, test_loader, maxlen, gt, device, None, attn, vis,
                                          ('Normal', 'normal'), True, log, batch_chunks, lanes)
    test.last_result = ucf_test.last_result
    return ret


def xd_test(args, model, test_loader, maxlen, prompt_text, gt, device, label_map, vis=False, attn=False, *, log=None,
            batch_chunks: Optional[int] = None, lanes: int = 1):
    """Drop-in for `train/xd_test.py`'s `test` (xd_test.py:15-26; call site xd_train.py:102-112): `label_map` is the
    8th positional, then (vis, attn) -- the reverse of ucf_test's order; every video's class is
    `label_map[cls.split('-')[0]]` whatever args.dataset says (xd_test.py:68, unconditional); Ano-AUC excludes
    'normal' only (xd_test.py:334)."""
    if label_map is None:
        raise TypeError("xd_test: label_map is required (xd_test.py:68 indexes it for every video)")
    ret, xd_test.last_result = _run_test(args, model, test_loader, maxlen, gt, device, _AlwaysRemap(label_map), attn, vis,
                                         ('normal',), False, log, batch_chunks, lanes)
    test.last_result = xd_test.last_result
    return ret


class _AlwaysRemap(dict):
    """Marks a label map whose remap applies for every dataset name (xd_test.py:68) -- `harness.test` applies it for
    args.dataset == 'xd' only, as root test.py:80-81 does."""


# ------------------------------------------------------------------------------------------------
# robustness sweep -- the second inference caller of the forward (/root/reference/test2.py:35-123, levels :29-32)
# ------------------------------------------------------------------------------------------------
SWEEP_LEVELS = (0, 0.05, 0.1, 0.2, 0.3, 0.5)      # fraction of a chunk's time steps that gets attenuated


def sweep_plan():
    """(name, sigma_img, sigma_ev) for the twelve runs of the reference's sweep: every level on the image features
    with the event features untouched, then the other way round."""
    return ([("IMG_NOISE", s, 0) for s in SWEEP_LEVELS] + [("EV_NOISE", 0, s) for s in SWEEP_LEVELS])


class SweepResult(tuple):
    """The 12 numbers `run_test` returns (test2.py:119-123), in its order, with names."""
    FIELDS = ("brier", "kl", "w_img_mean", "w_ev_mean", "auc", "ap", "w_img_anomalous", "w_ev_anomalous",
              "w_img_normal", "w_ev_normal", "w_img_change", "w_ev_change")

    def __getattr__(self, name):
        try:
            return self[self.FIELDS.index(name)]
        except ValueError:
            raise AttributeError(name) from None


class PerturbationSweep:
    """Robustness sweep over one test list, organised for the device instead of per video:

      * the videos are unpacked once (shape rule and unconditional `nan_to_num` of test2.py:52-60), packed across videos into
        batches of >= `batch_chunks` chunks (chunks are independent batch rows) and UPLOADED ONCE: the packed clean features stay
        resident on the device for all twelve levels;
      * the CLEAN pass -- identical for all twelve levels -- runs once; what later levels need from it stays on the
        device: the clean probabilities, and the per-dimension sums of the clean fusion weights;
      * a level draws its attenuated time steps with `torch.randperm(T)` once per video and modality, image first, in
        list order -- the draw sequence of test2.py:71-77, so seeding torch reproduces the reference's subsets -- and turns
        them into one fp32 row-scale vector per batch and modality (0.01 at the drawn time steps of every chunk of the video, as
        `x[:, idx]` does, 1 elsewhere) that the library applies in its input load (`iefvad_forward_scaled`): the clean
        features are never copied or modified;
      * the per-snippet means of the fusion weights come from the kernels (`w_i_mean`, `w_e_mean`); the full `w_i` / `w_e`
        are read only for the per-dimension sums of `w_img_change` / `w_ev_change` (test2.py:86-87);
      * Brier score, KL divergence, ROC-AUC / AP (`iefvad_auc_ap`) and the weight statistics are reduced on the device
        from per-snippet values and the 16-frames-per-snippet ground truth, never materialising the x16 repeat.

    On a HIP device `model` should be built with outputs="weights" (or "full").  With a CPU `device` (plumbing runs with a CPU
    model) the scale is applied to a copy with torch ops and the metrics come from sklearn as in test2.py:105-106."""

    def __init__(self, args, model, loader, gt, device, batch_chunks: int = 64, repeat: int = 16):
        self.model, self.device, self.repeat = model, torch.device(device), repeat
        self.T = T = args.visual_length
        videos: List[Tuple[torch.Tensor, torch.Tensor, int]] = []
        for item in loader:
            img, ev, n = item[0].squeeze(0), item[1].squeeze(0), int(item[3])
            if n < T:
                img, ev = img.unsqueeze(0), ev.unsqueeze(0)
            videos.append((torch.nan_to_num(img), torch.nan_to_num(ev), n))
        self.lengths = [n for _, _, n in videos]
        self.nchunks = [int(v[0].shape[0]) for v in videos]
        self.batches: List[List[int]] = [[]]
        load = 0
        for v, nch in enumerate(self.nchunks):
            self.batches[-1].append(v)
            load += nch
            if load >= batch_chunks and v + 1 < len(videos):
                self.batches.append([])
                load = 0
        # packed batches, resident on the device: (img [C, T, D], ev [C, T, D], valid row indices)
        self.packed = []
        for batch in self.batches:
            dts = {videos[v][m].dtype for v in batch for m in (0, 1)}
            dt = torch.float32 if len(dts) > 1 else next(iter(dts))
            img = torch.cat([videos[v][0].to(dt) for v in batch]).to(self.device)
            ev = torch.cat([videos[v][1].to(dt) for v in batch]).to(self.device)
            valid, off = [], 0
            for v in batch:
                valid.append(torch.arange(off * T, off * T + self.lengths[v]))
                off += self.nchunks[v]
            self.packed.append((img, ev, torch.cat(valid).to(self.device)))
        self.total = sum(self.lengths)
        g = torch.as_tensor(np.asarray(gt)[: repeat * self.total], dtype=torch.float64).reshape(self.total, repeat)
        self.gt = g.to(self.device)
        self.pos = self.gt.sum(dim=1)                        # anomalous frames of each snippet
        self.clean_passes = 0
        self._clean = None

    def _scales(self, batch, draws):
        """Row-scale vectors [C * T] (image, event) of one packed batch for one level's draws; None for an untouched modality."""
        out = []
        C = sum(self.nchunks[v] for v in batch)
        for m in (0, 1):
            if all(draws[v][m] is None or draws[v][m].numel() == 0 for v in batch):
                out.append(None)
                continue
            sc = torch.ones(C, self.T)
            off = 0
            for v in batch:
                idx = draws[v][m]
                if idx is not None and idx.numel():
                    sc[off:off + self.nchunks[v], idx] = 0.01
                off += self.nchunks[v]
            out.append(sc.reshape(-1).to(self.device))
        return out

    # one packed pass over the list; `draws[v]` = (image time steps, event time steps) to attenuate in video v, or None
    def _pass(self, draws=None):
        probs, wi_rows, we_rows = [], [], []
        wi_dim = we_dim = None
        on_gpu = self.device.type == 'cuda'
        with torch.no_grad():
            for batch, (img, ev, valid) in zip(self.batches, self.packed):
                if draws is None:
                    out = self.model(img, ev, None, None, None)
                else:
                    si, se = self._scales(batch, draws)
                    if on_gpu:
                        out = self.model(img, ev, None, None, None, row_scale=(si, se))
                    else:       # CPU plumbing: the product as torch forms it on a copy (test2.py:70-77)
                        xi = img if si is None else (img * si.reshape(-1, self.T, 1).to(img.dtype))
                        xe = ev if se is None else (ev * se.reshape(-1, self.T, 1).to(ev.dtype))
                        out = self.model(xi, xe, None, None, None)
                D = out['w_i'].shape[-1]
                probs.append(torch.sigmoid(out['logits'].reshape(-1)[valid]).float())
                wi = out['w_i'].reshape(-1, D)[valid]
                we = out['w_e'].reshape(-1, D)[valid]
                wi_rows.append(out['w_i_mean'].reshape(-1)[valid] if 'w_i_mean' in out else wi.float().mean(dim=1))
                we_rows.append(out['w_e_mean'].reshape(-1)[valid] if 'w_e_mean' in out else we.float().mean(dim=1))
                si, se = wi.sum(dim=0, dtype=torch.float64), we.sum(dim=0, dtype=torch.float64)
                wi_dim = si if wi_dim is None else wi_dim + si
                we_dim = se if we_dim is None else we_dim + se
        return {"p": torch.cat(probs), "wi_row": torch.cat(wi_rows), "we_row": torch.cat(we_rows),
                "wi_dim": wi_dim / self.total, "we_dim": we_dim / self.total}

    def clean(self):
        if self._clean is None:
            self._clean = self._pass()
            self.clean_passes += 1
        return self._clean

    def _auc_ap(self, p):
        if self.device.type == 'cuda':
            return device_auc_ap(p, self.gt, self.repeat)
        from sklearn.metrics import average_precision_score, roc_auc_score      # test2.py:105-106
        y = np.repeat(p.cpu().numpy(), self.repeat)
        g = self.gt.reshape(-1).cpu().numpy()
        return roc_auc_score(g, y), average_precision_score(g, y)

    def level(self, sigma_img=0, sigma_ev=0) -> SweepResult:
        c = self.clean()
        k_img, k_ev = int(self.T * sigma_img), int(self.T * sigma_ev)
        draws = []
        for _ in self.lengths:       # the reference's draw order: per video, image then event; no draw for a zero sigma
            di = torch.randperm(self.T)[:k_img] if sigma_img else None
            de = torch.randperm(self.T)[:k_ev] if sigma_ev else None
            draws.append((di, de))
        n = self._pass(draws)
        rep = self.repeat
        yc, yn = c["p"].double(), n["p"].double()
        brier = ((yn[:, None] - self.gt) ** 2).mean()
        eps = 1e-8
        pc, qn = yc.clamp(eps, 1 - eps), yn.clamp(eps, 1 - eps)
        kl = (pc * torch.log(pc / qn) + (1 - pc) * torch.log((1 - pc) / (1 - qn))).mean()
        auc, ap = self._auc_ap(n["p"])
        wi, we = n["wi_row"].double(), n["we_row"].double()
        P, N = self.pos.sum(), (rep - self.pos).sum()
        stats = torch.stack([brier, kl, wi.mean(), we.mean(), (wi * self.pos).sum() / P, (we * self.pos).sum() / P,
                             (wi * (rep - self.pos)).sum() / N, (we * (rep - self.pos)).sum() / N]).cpu().tolist()
        return SweepResult((stats[0], stats[1], stats[2], stats[3], auc, ap, stats[4], stats[5], stats[6], stats[7],
                            (c["wi_dim"] - n["wi_dim"]).float().cpu(), (c["we_dim"] - n["we_dim"]).float().cpu()))


def run_perturbation_test(args, model, loader, gt, device, sigma_img=0, sigma_ev=0, clean_cache: Optional[dict] = None,
                          batch_chunks: int = 64) -> SweepResult:
    """One level of the sweep with the call shape of the reference's `run_test(args, model, loader, gt, device,
    sigma_img, sigma_ev)` (test2.py:35) and its 12-tuple.  `clean_cache` (a dict the caller keeps across levels) holds
    the `PerturbationSweep`, so the unpacked list and the clean pass are shared by all levels."""
    model.eval()
    sweep = clean_cache.get("sweep") if clean_cache is not None else None
    if sweep is None:
        sweep = PerturbationSweep(args, model, loader, gt, device, batch_chunks)
        if clean_cache is not None:
            clean_cache["sweep"] = sweep
    return sweep.level(sigma_img, sigma_ev)


# ------------------------------------------------------------------------------------------------
# multi-GPU: contiguous, snippet-balanced shards + one ordered score gather
# ------------------------------------------------------------------------------------------------
def partition_by_snippets(lengths: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Cut the ordered test list into `world` contiguous [begin, end) video ranges with ~equal snippet
    counts.  Contiguity keeps rank-order concatenation equal to the reference's sequential order, on
    which the gt offsets depend (test.py:129,153)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    total = int(lengths.sum())
    cum = np.concatenate([[0], np.cumsum(lengths)])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        j = int(np.searchsorted(cum, target, side='left'))
        if j > 0 and abs(cum[j - 1] - target) <= abs(cum[min(j, len(cum) - 1)] - target):
            j -= 1
        j = min(max(j, cuts[-1]), len(lengths))
        cuts.append(j)
    cuts.append(len(lengths))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class ScoreComm:
    """The RCCL communicator of the score gather, owned by libiefvad (`iefvad_comm_*`, include/iefvad.h): created
    once per process over an initialised torch.distributed group, whose store only carries the 128-byte unique id
    from rank 0 to the other ranks.  The gather itself is `iefvad_gather_scores` on the caller's HIP stream."""

    CREATE_TIMEOUT_S = 120.0      # iefvad_comm_create is a collective (ncclCommInitRank): a rank that never arrives must not hang the job

    def __init__(self, device, group=None, create_timeout: Optional[float] = None, _hooks: Optional[dict] = None):
        """`_hooks` (tests only: tests/test_bench_launcher_cpu.py rehearses the failure paths on CPU ranks) replaces the four
        library calls of the handshake -- {'version', 'unique_id', 'create', 'destroy', 'nranks'} -- with callables."""
        import ctypes as C
        import threading
        import torch.distributed as dist
        from . import lib as _lib
        self._lib = _lib.load_library()
        self._last_error = _lib.last_error
        self._hooks = _hooks or {}
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        timeout = self.CREATE_TIMEOUT_S if create_timeout is None else float(create_timeout)
        # every failure below is made SYMMETRIC: a rank that raised alone would leave the others blocked in a collective.
        # First of all: can EVERY rank bind librccl?  (iefvad_comm_create is itself a collective -- ncclCommInitRank -- so a
        # rank that cannot even dlopen the library must be found out before anybody enters it.)
        version = self._hooks.get('version', lambda: int(self._lib.iefvad_rccl_version()))
        ver = torch.tensor([int(version())], dtype=torch.int64, device=self.device)
        dist.all_reduce(ver, op=dist.ReduceOp.MIN, group=group)
        if int(ver.item()) <= 0:
            raise RuntimeError("librccl cannot be bound on at least one rank (iefvad_rccl_version() == 0)")
        self.rccl_version = int(ver.item())
        ident = [None]
        if self.rank == 0:
            if 'unique_id' in self._hooks:
                ident = [self._hooks['unique_id']()]
            else:
                buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
                if self._lib.iefvad_comm_unique_id(buf) != 0:
                    ident = ["iefvad_comm_unique_id: " + self._last_error()]      # a str instead of the id: everybody raises
                else:
                    ident = [bytes(buf.raw)]
        dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not isinstance(ident[0], bytes):
            raise RuntimeError(str(ident[0]))
        self._h = C.c_void_p()
        # the collective create runs on a helper thread and is waited for with a deadline: if a peer died or never enters
        # ncclCommInitRank, this rank gives up after `timeout` seconds, reports failure in the all-reduce below -- which its
        # peers reach the same way -- and the job continues on torch.distributed's transport instead of hanging
        box = {}

        def create():
            try:
                if 'create' in self._hooks:
                    box['h'] = self._hooks['create'](ident[0], self.world, self.rank)
                    box['rc'] = 0
                else:
                    with torch.cuda.device(self.device):
                        box['rc'] = self._lib.iefvad_comm_create(C.c_char_p(ident[0]), self.world, self.rank, C.byref(self._h))
                    box['err'] = "" if box['rc'] == 0 else "iefvad_comm_create: " + self._last_error()     # thread-local message: read it here
            except Exception as e:       # noqa: BLE001 -- reported through the all-reduce like any other failure
                box['rc'], box['err'] = 1, f"iefvad_comm_create: {e}"

        th = threading.Thread(target=create, name="iefvad_comm_create", daemon=True)
        th.start()
        th.join(timeout)
        if th.is_alive():
            rc, err = 1, f"iefvad_comm_create did not return within {timeout:.0f} s (a peer never entered the collective)"
            self._abandoned = th       # still inside the collective: its handle is never used, the daemon thread dies with the process
        else:
            rc, err = box.get('rc', 1), box.get('err', "")
            if rc == 0 and 'h' in box:
                self._h = box['h']
        bad = torch.tensor([1 if rc != 0 else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        if int(bad.item()) != 0:
            if rc == 0:
                self.close()
            self._h = None
            raise RuntimeError(err or "iefvad_comm_create failed on another rank")
        self.nranks = int(self._hooks['nranks'](self._h)) if 'nranks' in self._hooks else int(self._lib.iefvad_comm_nranks(self._h))      # what RCCL reports

    def gather(self, local: torch.Tensor, counts: Optional[Sequence[int]] = None) -> torch.Tensor:
        import ctypes as C
        local = local.reshape(-1).float().contiguous()
        assert local.is_cuda and local.device == self.device
        if counts is None:
            counts = [local.numel()] * self.world
        assert len(counts) == self.world and counts[self.rank] == local.numel(), (counts, local.numel())
        out = torch.empty(int(sum(counts)), dtype=torch.float32, device=self.device)
        carr = (C.c_int64 * self.world)(*[int(c) for c in counts])
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            rc = self._lib.iefvad_gather_scores(self._h, C.c_void_p(local.data_ptr()), local.numel(), carr,
                                                C.c_void_p(out.data_ptr()), out.numel(), st)
        if rc != 0:
            raise RuntimeError("iefvad_gather_scores: " + self._last_error())
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if 'destroy' in getattr(self, "_hooks", {}):
                self._hooks['destroy'](self._h)
            else:
                self._lib.iefvad_comm_destroy(self._h)
            self._h = None


_score_comms: Dict[object, ScoreComm] = {}


def gather_scores(local: torch.Tensor, group=None, counts: Optional[Sequence[int]] = None,
                  use_library: bool = True) -> torch.Tensor:
    """Rank-order concatenation of every rank's fp32 score vector, on every rank (= the reference's sequential
    order, test.py:123-129,153, because shards are contiguous ranges of the test list).

    `counts` are the per-rank lengths.  Every rank can compute them from the shared test list
    (`partition_by_snippets`), so pass them whenever possible: then nothing but the scores travels and nothing
    synchronises with the host -- equal counts are ONE all-gather.  Without `counts` the lengths are exchanged first
    (one small all-gather and a host read).
    HIP tensors on an "nccl" (= RCCL) group go through the library's own `iefvad_gather_scores`; CPU tensors (the
    gloo tests), and every tensor when `use_library` is False, through torch.distributed (`all_gather_into_tensor`,
    padded to the longest shard when the counts differ)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    local = local.reshape(-1).float()
    if counts is None:
        n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
        allc = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(allc, n, group=group)
        counts = allc.tolist()
    counts = [int(c) for c in counts]
    if use_library and local.is_cuda and dist.get_backend(group) == "nccl":
        key = (group, local.device.index)
        if key not in _score_comms:
            _score_comms[key] = ScoreComm(local.device, group)
        return _score_comms[key].gather(local, counts)
    if len(set(counts)) == 1:
        out = torch.empty(world * counts[0], dtype=torch.float32, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(counts)
    buf = torch.zeros(mx, dtype=torch.float32, device=local.device)
    buf[:local.numel()] = local
    out = torch.empty(world * mx, dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return torch.cat([out[r * mx:r * mx + counts[r]] for r in range(world)])


class ScoreGatherer:
    """How a multi-rank job gathers its score vectors, decided ONCE and by all ranks together: the library's RCCL gather
    (`ScoreComm` -> `iefvad_gather_scores`) when every rank can create the communicator -- `ScoreComm.__init__` fails on
    all ranks or on none -- else torch.distributed's `all_gather_into_tensor` (`gather_scores(use_library=False)`).
    `label` says which one runs.  There is no switch in mid-run: an enqueue error of the library gather on one rank would
    leave its peers inside the collective, so it propagates and ends the job instead of being papered over."""

    LIB = "iefvad_gather_scores (libiefvad -> librccl: ncclAllGather, or grouped ncclSend/ncclRecv for unequal shards, on the forward's stream)"
    TORCH = "torch.distributed all_gather_into_tensor"

    def __init__(self, device, group=None, prefer_library: bool = True, create_timeout: Optional[float] = None, _hooks=None):
        self.group, self.comm = group, None
        if prefer_library:
            try:
                self.comm = ScoreComm(device, group, create_timeout, _hooks)
                self.label = self.LIB
            except Exception as e:                  # symmetric (see ScoreComm.__init__): every rank lands here or none
                self.label = f"{self.TORCH} (library gather unavailable: {e})"
        else:
            self.label = self.TORCH

    def __call__(self, scores: torch.Tensor, counts: Optional[Sequence[int]] = None) -> torch.Tensor:
        if self.comm is not None:
            return self.comm.gather(scores, counts)
        return gather_scores(scores, self.group, counts, use_library=False)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


def shard_counts(lengths: Sequence[int], world: int) -> List[int]:
    """Snippets per rank under `partition_by_snippets` -- what every rank passes to `gather_scores(counts=...)`."""
    lengths = np.asarray(lengths, dtype=np.int64)
    return [int(lengths[a:b].sum()) for a, b in partition_by_snippets(lengths, world)]
