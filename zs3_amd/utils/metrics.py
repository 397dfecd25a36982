"""Segmentation metrics with the Evaluator surface of zs3/utils/metrics.py:4-82 (same attribute and method names, same
return tuples) -- plus a device path: `add_batch` also takes GPU tensors, and
`add_batch_logits` counts straight from the network's logits (argmax fused, and for low-resolution logits the final
bilinear upsample fused as well: zs3_argmax_confusion), so that validation (train_pascal.py:130-134) never ships the
[B, C, 513, 513] output to the host.  Counts are integers: the device path is bit-identical to the numpy path."""
import numpy as np
import torch


def _safe_div(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return a / b


class Evaluator:
    def __init__(self, num_class, seen_classes_idx=None, unseen_classes_idx=None):
        self.num_class = num_class
        self.seen_classes_idx = seen_classes_idx
        self.unseen_classes_idx = unseen_classes_idx
        self._host = np.zeros((num_class, num_class))
        self._dev = None    # int64 [C, C] counters on the GPU, folded into the host matrix when it is read

    # ---- the confusion matrix (rows = ground truth, columns = prediction), float64 like the reference's
    @property
    def confusion_matrix(self):
        if self._dev is not None:
            self._host = self._host + self._dev.cpu().numpy().astype(np.float64)
            self._dev.zero_()
        return self._host

    @confusion_matrix.setter
    def confusion_matrix(self, value):
        self._host = np.asarray(value, dtype=np.float64)
        if self._dev is not None:
            self._dev.zero_()

    def _split(self):
        return bool(self.seen_classes_idx) and bool(self.unseen_classes_idx)

    # ---- metrics (metrics.py:11-71)
    def Pixel_Accuracy(self):
        cm = self.confusion_matrix
        diag = np.diag(cm)
        acc = _safe_div(diag.sum(), cm.sum())
        if self._split():
            s, u = self.seen_classes_idx, self.unseen_classes_idx
            return acc, _safe_div(diag[s].sum(), cm[s, :].sum()), _safe_div(diag[u].sum(), cm[u, :].sum())
        return acc

    def Pixel_Accuracy_Class(self):
        cm = self.confusion_matrix
        by_class = _safe_div(np.diag(cm), cm.sum(axis=1))
        acc = np.nanmean(np.nan_to_num(by_class))
        if self._split():
            return (acc, by_class, np.nanmean(np.nan_to_num(by_class[self.seen_classes_idx])),
                    np.nanmean(np.nan_to_num(by_class[self.unseen_classes_idx])))
        return acc, by_class

    def _iou(self):
        cm = self.confusion_matrix
        diag = np.diag(cm)
        return _safe_div(diag, cm.sum(axis=1) + cm.sum(axis=0) - diag)

    def Mean_Intersection_over_Union(self):
        iou = self._iou()
        miou = np.nanmean(np.nan_to_num(iou))
        if self._split():
            return (miou, iou, np.nanmean(np.nan_to_num(iou[self.seen_classes_idx])),
                    np.nanmean(np.nan_to_num(iou[self.unseen_classes_idx])))
        return miou, iou

    def Frequency_Weighted_Intersection_over_Union(self):
        cm = self.confusion_matrix
        freq = _safe_div(cm.sum(axis=1), cm.sum())
        iou = self._iou()
        fw = (freq[freq > 0] * iou[freq > 0]).sum()
        if self._split():
            out = [fw]
            for idx in (self.seen_classes_idx, self.unseen_classes_idx):
                f, i = freq[idx], iou[idx]
                out.append((f[f > 0] * i[f > 0]).sum())
            return tuple(out)
        return fw

    # ---- accumulation
    def _generate_matrix(self, gt_image, pre_image):
        gt_image, pre_image = np.asarray(gt_image), np.asarray(pre_image)
        keep = (gt_image >= 0) & (gt_image < self.num_class)
        label = self.num_class * gt_image[keep].astype("int") + pre_image[keep]
        return np.bincount(label, minlength=self.num_class ** 2).reshape(self.num_class, self.num_class)

    def _device_counters(self, device):
        if self._dev is None or self._dev.device != device:
            if self._dev is not None:
                _ = self.confusion_matrix   # fold the old device's counts in
            self._dev = torch.zeros((self.num_class, self.num_class), dtype=torch.int64, device=device)
        return self._dev

    def add_batch(self, gt_image, pre_image):
        """numpy arrays like the reference, or torch tensors (CUDA tensors are counted on the device)."""
        assert tuple(gt_image.shape) == tuple(pre_image.shape)
        if torch.is_tensor(gt_image) and gt_image.is_cuda:
            conf = self._device_counters(gt_image.device)
            gt = gt_image.reshape(-1)
            keep = (gt >= 0) & (gt < self.num_class)
            label = self.num_class * gt[keep].long() + pre_image.reshape(-1)[keep].long()
            conf += torch.bincount(label, minlength=self.num_class ** 2).view(self.num_class, self.num_class)
            return
        if torch.is_tensor(gt_image):
            gt_image, pre_image = gt_image.numpy(), pre_image.numpy()
        self._host = self.confusion_matrix + self._generate_matrix(gt_image, pre_image)

    def add_batch_logits(self, gt_image, logits):
        """gt_image: [B, H, W] labels (float or int64); logits: [B, C, h, w] network output on the GPU -- full resolution
        (`model(image)`) or the low-resolution class scores before the final upsample (h, w) != (H, W), which are then
        resized with align_corners=True on the fly.  argmax + histogram happen in one kernel."""
        from .. import ops
        from .._lib import I, P, check, lib, require_gpu, stream
        require_gpu(gt_image, logits)
        b, c, h, w = logits.shape
        assert c == self.num_class and gt_image.shape[0] == b and gt_image.dim() == 3
        x = ops.nhwc(logits)
        x = x if x.stride(-1) == 1 else x.contiguous()
        ld = ops._check_nhwc(x)
        gt = gt_image.contiguous()
        if gt.dtype not in (torch.float32, torch.int64):
            gt = gt.long()
        conf = self._device_counters(logits.device)
        check(lib().zs3_argmax_confusion(P(x), I(ld), I(b), I(h), I(w), I(c), P(gt), I(int(gt.dtype == torch.int64)),
                                         I(gt.shape[1]), I(gt.shape[2]), P(conf), stream()), "zs3_argmax_confusion")

    def reset(self):
        self._host = np.zeros((self.num_class,) * 2)
        if self._dev is not None:
            self._dev.zero_()


class Evaluator_seen_unseen:
    """Score tuples of zs3/utils/metrics.py:88-196 (`eval_pascal.py:83` builds one): overall / seen-rows / unseen-rows /
    per-class-rows (accuracy, mean class accuracy, mean IU, frequency-weighted IU).  The reference histograms every image
    up to 2 + num_class times with different ground-truth masks; a masked histogram is the full confusion matrix with the
    other ground-truth rows zeroed, so one histogram per image is enough here and the metrics are identical."""

    def __init__(self, num_class, unseen_classes_idx):
        self.num_class = num_class
        self.unseen_classes_idx = unseen_classes_idx

    def _fast_hist(self, label_true, label_pred, n_class, target="all", unseen=None):
        label_true, label_pred = np.asarray(label_true), np.asarray(label_pred)
        keep = (label_true >= 0) & (label_true < n_class)
        hist = np.bincount(n_class * label_true[keep].astype(int) + label_pred[keep], minlength=n_class ** 2)
        return self._rows(hist.reshape(n_class, n_class), target, unseen)

    def _fast_hist_specific_class(self, label_true, label_pred, n_class, target_class):
        return self._rows(self._fast_hist(label_true, label_pred, n_class), "class", target_class)

    @staticmethod
    def _rows(hist, target, which):
        """ground-truth rows of `hist` selected by the reference's `target` modes"""
        if target == "all":
            return hist
        n = hist.shape[0]
        sel = np.zeros(n, dtype=bool)
        sel[np.atleast_1d(np.asarray(which, dtype=int))] = True
        if target == "seen":
            sel = ~sel
        return hist * sel[:, None]

    def _hist_to_metrics(self, hist):
        hist = np.asarray(hist, dtype=np.float64)
        diag, rows, total = np.diag(hist), hist.sum(axis=1), hist.sum()
        acc = 0.0 if total == 0 else diag.sum() / total
        acc_cls = np.nanmean(_safe_div(diag, rows))
        iu = _safe_div(diag, rows + hist.sum(axis=0) - diag)
        freq = _safe_div(rows, total)
        return acc, acc_cls, np.nanmean(iu), (freq[freq > 0] * iu[freq > 0]).sum()

    def label_accuracy_score(self, label_trues, label_preds, by_class=False):
        n = self.num_class
        hist = np.zeros((n, n))
        class_hist = [np.zeros((n, n)) for _ in range(n)] if by_class else None
        for lt, lp in zip(label_trues, label_preds):
            lt, lp = np.asarray(lt).flatten(), np.asarray(lp).flatten()
            h = self._fast_hist(lt, lp, n)
            hist += h
            if by_class:   # the reference adds a class's rows only for images that contain the class: the same rows
                for c in np.unique(lt).astype(np.int32):
                    if c != 255:
                        class_hist[c] += self._rows(h, "class", c)
        metrics = self._hist_to_metrics(hist)
        if self.unseen_classes_idx:
            metrics = (metrics, self._hist_to_metrics(self._rows(hist, "seen", self.unseen_classes_idx)),
                       self._hist_to_metrics(self._rows(hist, "unseen", self.unseen_classes_idx)))
        if by_class:
            return metrics, [self._hist_to_metrics(h) for h in class_hist]
        return metrics
