"""Host-side callers of the forward: chunker, feature-file dataset, the evaluation loop and its
multi-GPU sharding.  Counterparts of
  * `process_split` / `pad`                       /root/reference/data/tools.py:81-86,100-114
  * `UCF_Dataset` / `XD_Dataset` / `Shang_Dataset` /root/reference/data/dataset.py:8-127 (test mode)
  * `test()`                                      /root/reference/test.py:46-212,
                                                   train/ucf_test.py:16-216, train/xd_test.py:15-210
  * `compute_ano_auc`                             /root/reference/test.py:332-348
  * `run_test` (robustness sweep)                 /root/reference/test2.py:35-123
The model is any callable with the reference's `model(img, ev, padding_mask, text, lengths)` contract.
"""
from __future__ import annotations

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
                    verbose: bool = True, normal_keys=('Normal',)) -> Dict[str, object]:
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
    per_class = {}
    for c in keys:
        cls_pred = np.concatenate(cw_pred[c])   # raises on an empty class exactly as test.py:166-167 does
        cls_gt = np.concatenate(cw_gt[c])
        if len(cls_gt) == 0 or sum(cls_gt) == 0:
            continue
        c_roc = roc_auc_score(cls_gt, np.repeat(cls_pred, 16))
        c_ap = average_precision_score(cls_gt, np.repeat(cls_pred, 16))
        if verbose:
            print(c, 'ROC: {:.2f}  AP: {:.2f}'.format(c_roc * 100, c_ap * 100))
        per_class[c] = (c_roc, c_ap)
    if verbose:
        print('-------------------------------------------------')
    return {"roc": roc, "ap": ap, "ano_auc": ano, "per_class": per_class}


def device_auc_ap(scores: torch.Tensor, gt: torch.Tensor, repeat: int = 16) -> Tuple[float, float]:
    """ROC-AUC and average precision of `np.repeat(scores, repeat)` against frame-level `gt` without leaving
    the device and without materialising the repeat (the metric tail of test.py:158-159 costs sklearn a sort
    of 16 x N points on the host).  Ties are handled as sklearn does: thresholds are the DISTINCT score
    values; a snippet contributes its `repeat` frames at one threshold.
      AUC = (sum over thresholds of trapezoids) = [sum_g P_g * (N_below_g + N_g / 2)] / (P * N)
      AP  = sum_g (R_g - R_{g-1}) * Prec_g   with groups g in decreasing score order."""
    s = scores.reshape(-1).to(torch.float64)
    g = gt.reshape(-1, repeat).to(torch.float64).to(s.device)
    assert g.shape[0] == s.shape[0], "gt must hold `repeat` frames per snippet"
    pos = g.sum(dim=1)                        # positives among this snippet's frames
    neg = repeat - pos
    order = torch.argsort(s, descending=True, stable=True)
    s, pos, neg = s[order], pos[order], neg[order]
    new_group = torch.ones_like(s, dtype=torch.bool)
    new_group[1:] = s[1:] != s[:-1]
    gid = torch.cumsum(new_group.to(torch.int64), 0) - 1
    ng = int(gid[-1].item()) + 1
    Pg = torch.zeros(ng, dtype=torch.float64, device=s.device).index_add_(0, gid, pos)
    Ng = torch.zeros(ng, dtype=torch.float64, device=s.device).index_add_(0, gid, neg)
    P, N = Pg.sum(), Ng.sum()
    tp = torch.cumsum(Pg, 0)
    fp = torch.cumsum(Ng, 0)
    # AUC: positives of group g beat every negative in later (lower-score) groups and tie with their own
    neg_below = N - fp
    auc = ((Pg * (neg_below + 0.5 * Ng)).sum() / (P * N)).item()
    prec = tp / (tp + fp)
    ap = ((Pg / P) * prec).sum().item()
    return auc, ap


# ------------------------------------------------------------------------------------------------
# the evaluation loop
# ------------------------------------------------------------------------------------------------
def _unpack_item(item, maxlen, dataset, label_map):
    """Shape rule of test.py:77-88 applied to one DataLoader item (batch_size=1)."""
    img = item[0].squeeze(0)
    ev = item[1].squeeze(0)
    cls = item[2][0] if isinstance(item[2], (list, tuple)) else item[2]
    if dataset == 'xd' and label_map is not None:
        cls = label_map[cls.split('-')[0]]            # xd_test.py:68
    length = int(item[3])
    if length < maxlen:
        img = img.unsqueeze(0)
        ev = ev.unsqueeze(0)
    if torch.isnan(img).any():                        # conditional nan_to_num, test.py:90-95
        img = torch.nan_to_num(img, nan=0.0)
    if torch.isnan(ev).any():
        ev = torch.nan_to_num(ev, nan=0.0)
    return img, ev, cls, length


def score_loader(model: Callable, test_loader: Iterable, maxlen: int, device, dataset: str = 'ucfcrime',
                 label_map=None, batch_chunks: int = 0, skip_empty_chunks: bool = True):
    """Per-video sigmoid scores and mean fusion weights, in loader order.

    batch_chunks == 0: one forward per video with B = that video's chunk count -- the reference's call
    pattern (test.py:76-117).  batch_chunks > 0: chunks of consecutive videos are packed into one
    forward of up to `batch_chunks` chunks; legal because chunks are independent batch rows
    (imf_vad.py:115 attends within a chunk), and the trailing all-zero chunk of a len % 256 == 0 video,
    whose rows the reference slices away (test.py:121), is not computed when `skip_empty_chunks`."""
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

    def flush():
        nonlocal pend, pend_chunks, total
        if not pend:
            return
        # dtype as the files hold it (dataset.py:37-38,49-50); the model widens with `.to(torch.float)` per modality
        # (imf_vad.py:41-42), so a batch that mixes dtypes -- across videos OR between the two modalities -- is widened
        # to fp32 here rather than narrowed to the first tensor's type
        dts = {p[0].dtype for p in pend} | {p[1].dtype for p in pend}
        dt = torch.float32 if len(dts) > 1 else pend[0][0].dtype
        img = torch.cat([p[0].to(dt) for p in pend], dim=0).to(device, non_blocking=True)
        ev = torch.cat([p[1].to(dt) for p in pend], dim=0).to(device, non_blocking=True)
        out = model(img, ev, None, None, None)
        logits = out['logits'].reshape(-1)
        if 'w_i_mean' in out:
            wi, we = out['w_i_mean'].reshape(-1), out['w_e_mean'].reshape(-1)
        else:
            wi = out['w_i'].reshape(-1, out['w_i'].shape[-1]).mean(dim=-1)     # test.py:131-136
            we = out['w_e'].reshape(-1, out['w_e'].shape[-1]).mean(dim=-1)
        dev_prob.append(torch.sigmoid(logits).float())
        dev_wi.append(wi.float())
        dev_we.append(we.float())
        off = total
        for ci, _, n in pend:
            spans.append((off, n))                           # logits1[0:len_cur] -> sigmoid, test.py:119-121
            off += ci.shape[0] * maxlen
        total = off
        pend, pend_chunks = [], 0

    with torch.no_grad():
        for item in test_loader:
            img, ev, cls, n = _unpack_item(item, maxlen, dataset, label_map)
            classes.append(cls)
            if batch_chunks > 0 and skip_empty_chunks and n >= maxlen and n % maxlen == 0:
                img, ev = img[:-1], ev[:-1]               # the all-zero chunk (tools.py:105-112)
            pend.append((img, ev, n))
            pend_chunks += img.shape[0]
            if batch_chunks <= 0 or pend_chunks >= batch_chunks:
                flush()
        flush()
        if dev_prob:
            prob = torch.cat(dev_prob).cpu().numpy()
            wi = torch.cat(dev_wi).cpu().numpy()
            we = torch.cat(dev_we).cpu().numpy()
        else:
            prob = wi = we = np.zeros(0, np.float32)
    scores = [prob[o:o + n].copy() for o, n in spans]
    wi_means = [wi[o:o + n].copy() for o, n in spans]
    we_means = [we[o:o + n].copy() for o, n in spans]
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
                 batch_chunks: int = 64, workers: int = 2, prefetch: int = 2):
        self.paths, self.labels = list(paths), list(labels)
        self.clip_dim, self.event_dir, self.device = clip_dim, event_dir, torch.device(device)
        self.batch_chunks, self.workers, self.prefetch = batch_chunks, workers, max(1, prefetch)

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
    def _fill(dst: np.ndarray, src):
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
        if np.issubdtype(dst.dtype, np.floating) and np.isnan(dst).any():
            # conditional nan_to_num (test.py:90-95): NaN -> 0, +-inf -> the SOURCE dtype's max / min
            fi = np.finfo(dtype)
            np.nan_to_num(dst, copy=False, nan=0.0, posinf=float(fi.max), neginf=float(fi.min))

    @classmethod
    def _fill_many(cls, jobs):
        for dst, src in jobs:
            cls._fill(dst, src)

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
    pipe = FeatureFilePipeline(paths, labels, args.visual_length, EVENT_DIR[dataset], device, batch_chunks, workers)
    model.eval()
    t0 = time.perf_counter()
    outs, metas = [], []
    with torch.no_grad():
        for img, ev, meta in pipe.batches():
            o = model(img, ev, None, None, None)
            outs.append(o['logits'].reshape(img.shape[0], -1))
            metas.append(meta)
    T = args.visual_length
    pieces = []
    for lg, meta in zip(outs, metas):
        off = 0
        for _, n, nch in meta:
            pieces.append(lg[off:off + nch].reshape(-1)[:n])
            off += nch
    scores_dev = torch.sigmoid(torch.cat(pieces))
    if scores_dev.is_cuda:
        torch.cuda.synchronize(scores_dev.device)
    t1 = time.perf_counter()
    res: Dict[str, object] = {}
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


def test(args, model, test_loader, maxlen, prompt_text, gt, device, attn=False, vis=False, label_map=None,
         batch_chunks: int = 0, normal_keys=('Normal',)):
    """Same positional signature and return value as the reference's `test()` (test.py:46-56;
    ucf_test.py:16-26; xd_test.py passes `label_map` as an extra positional, :23).
    Returns (ROC1, AP1) -- or (ROC1, AP1, attn_weights, labels) when attn=True, where attn_weights is the
    empty list the reference also returns (it never fills it, test.py:73,209-210)."""
    model.to(device)
    model.eval()
    scores, classes, wi, we = score_loader(model, test_loader, maxlen, device, args.dataset, label_map, batch_chunks)
    res = evaluate_scores(scores, classes, gt, args.dataset, verbose=True, normal_keys=normal_keys)
    test.last_result = dict(res, scores=scores, classes=classes, w_i_mean=wi, w_e_mean=we)
    if vis:
        print("[iefvad_amd] vis=True: plotting (test.py:177-207) is outside the hot-path scope; skipped")
    if attn:
        return res["roc"], res["ap"], [], classes
    return res["roc"], res["ap"]


# ------------------------------------------------------------------------------------------------
# robustness sweep (second inference caller): /root/reference/test2.py:16-123
# ------------------------------------------------------------------------------------------------
SWEEP_CFGS = {   # test2.py:29-32
    "IMG_NOISE": {"sigma_img": [0, 0.05, 0.1, 0.2, 0.3, 0.5], "sigma_ev": [0]},
    "EV_NOISE": {"sigma_ev": [0, 0.05, 0.1, 0.2, 0.3, 0.5], "sigma_img": [0]},
}


def brier_score(pred, gt):
    return np.mean((pred - gt) ** 2)


def kl_divergence(pred_clean, pred_noisy, eps=1e-8):
    p, q = np.clip(pred_clean, eps, 1 - eps), np.clip(pred_noisy, eps, 1 - eps)
    return np.mean(p * np.log(p / q) + (1 - p) * np.log((1 - p) / (1 - q)))


def run_perturbation_test(args, model, loader, gt, device, sigma_img=0, sigma_ev=0, clean_cache: Optional[dict] = None):
    """Counterpart of `run_test` (test2.py:35-123): per video one clean and one perturbed forward, where
    the perturbation scales a random `int(T * sigma)` subset of TIME STEPS (dim 1, all chunks alike) of one
    modality by 0.01 (:71-77; indices from `torch.randperm`, so seeding torch reproduces the reference's
    draw sequence).  Returns the reference's 12-tuple.  `clean_cache` (a dict kept by the caller across
    sweep levels) lets the clean forward, identical for every level, run once instead of 12 times."""
    from sklearn.metrics import average_precision_score, roc_auc_score
    model.eval()
    maxlen = args.visual_length
    repeat = 16
    preds_clean, preds_noisy = [], []
    w_img_orig, w_ev_orig, w_img_all, w_ev_all = [], [], [], []

    def weights(out, length):
        wi = out['w_i'].reshape(-1, out['w_i'].shape[-1])[:length].float().cpu()
        we = out['w_e'].reshape(-1, out['w_e'].shape[-1])[:length].float().cpu()
        return wi, we

    with torch.no_grad():
        for vid, (visuals, events, _, length) in enumerate(loader):
            visuals = visuals.squeeze(0)
            events = events.squeeze(0)
            length = int(length)
            if length < maxlen:
                visuals = visuals.unsqueeze(0)
                events = events.unsqueeze(0)
            visuals = torch.nan_to_num(visuals).to(device)       # unconditional here (test2.py:59-60)
            events = torch.nan_to_num(events).to(device)

            if clean_cache is not None and vid in clean_cache:
                p_c, wi_c, we_c = clean_cache[vid]
            else:
                out_c = model(visuals, events, None, None, torch.tensor([length]))
                p_c = torch.sigmoid(out_c['logits'].reshape(-1)[:length]).float().cpu()
                wi_c, we_c = weights(out_c, length)
                if clean_cache is not None:
                    clean_cache[vid] = (p_c, wi_c, we_c)
            w_img_orig.append(wi_c)
            w_ev_orig.append(we_c)

            v_p, e_p = visuals.clone(), events.clone()
            if sigma_img:
                idx = torch.randperm(v_p.shape[1])[: int(v_p.shape[1] * sigma_img)]
                v_p[:, idx] = v_p[:, idx] * 0.01
            if sigma_ev:
                idx = torch.randperm(e_p.shape[1])[: int(e_p.shape[1] * sigma_ev)]
                e_p[:, idx] = e_p[:, idx] * 0.01
            out_n = model(v_p, e_p, None, None, torch.tensor([length]))
            p_n = torch.sigmoid(out_n['logits'].reshape(-1)[:length]).float().cpu()
            wi_n, we_n = weights(out_n, length)
            preds_clean.append(p_c)
            preds_noisy.append(p_n)
            w_img_all.append(wi_n)
            w_ev_all.append(we_n)

    yc = torch.cat(preds_clean).numpy()
    yn = torch.cat(preds_noisy).numpy()
    yc_rep, yn_rep = np.repeat(yc, repeat), np.repeat(yn, repeat)
    gt_slice = gt[: len(yn_rep)]
    w_img_all, w_ev_all = torch.cat(w_img_all), torch.cat(w_ev_all)
    w_img_orig, w_ev_orig = torch.cat(w_img_orig), torch.cat(w_ev_orig)
    w_img_change = w_img_orig.mean(0) - w_img_all.mean(0)
    w_ev_change = w_ev_orig.mean(0) - w_ev_all.mean(0)
    w_img = np.repeat(w_img_all.mean(1).numpy(), repeat)
    w_ev = np.repeat(w_ev_all.mean(1).numpy(), repeat)
    brier = brier_score(yn_rep, gt_slice)
    kl = kl_divergence(yc_rep, yn_rep)
    auc = roc_auc_score(gt_slice, yn_rep)
    ap = average_precision_score(gt_slice, yn_rep)
    return (brier, kl, w_img.mean(), w_ev.mean(), auc, ap, np.mean(w_img[gt_slice == 1]), np.mean(w_ev[gt_slice == 1]),
            np.mean(w_img[gt_slice == 0]), np.mean(w_ev[gt_slice == 0]), w_img_change, w_ev_change)


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

    def __init__(self, device, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import lib as _lib
        self._lib = _lib.load_library()
        self._last_error = _lib.last_error
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        ident = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
            if self._lib.iefvad_comm_unique_id(buf) != 0:
                raise RuntimeError("iefvad_comm_unique_id: " + self._last_error())
            ident = [bytes(buf.raw)]
        dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self._lib.iefvad_comm_create(C.c_char_p(ident[0]), self.world, self.rank, C.byref(self._h))
        if rc != 0:
            raise RuntimeError("iefvad_comm_create: " + self._last_error())
        self.nranks = int(self._lib.iefvad_comm_nranks(self._h))      # what RCCL reports

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
                                                C.c_void_p(out.data_ptr()), st)
        if rc != 0:
            raise RuntimeError("iefvad_gather_scores: " + self._last_error())
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.iefvad_comm_destroy(self._h)
            self._h = None


_score_comms: Dict[object, ScoreComm] = {}


def gather_scores(local: torch.Tensor, group=None, counts: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Rank-order concatenation of every rank's fp32 score vector, on every rank (= the reference's sequential
    order, test.py:123-129,153, because shards are contiguous ranges of the test list).

    `counts` are the per-rank lengths.  Every rank can compute them from the shared test list
    (`partition_by_snippets`), so pass them whenever possible: then nothing but the scores travels and nothing
    synchronises with the host -- equal counts are ONE all-gather.  Without `counts` the lengths are exchanged first
    (one small all-gather and a host read).
    HIP tensors on an "nccl" (= RCCL) group go through the library's own `iefvad_gather_scores`; CPU tensors (the
    gloo tests) through torch.distributed."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    local = local.reshape(-1).float()
    if counts is None:
        n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
        allc = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(allc, n, group=group)
        counts = allc.tolist()
    counts = [int(c) for c in counts]
    if local.is_cuda and dist.get_backend(group) == "nccl":
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


def shard_counts(lengths: Sequence[int], world: int) -> List[int]:
    """Snippets per rank under `partition_by_snippets` -- what every rank passes to `gather_scores(counts=...)`."""
    lengths = np.asarray(lengths, dtype=np.int64)
    return [int(lengths[a:b].sum()) for a, b in partition_by_snippets(lengths, world)]
