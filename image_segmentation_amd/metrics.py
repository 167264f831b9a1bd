"""MetricsHistory -- drop-in for the reference's utils/MetricsHistory.py:4-183.

Same constructor, accumulators (float64 on the CPU, MetricsHistory.py:21-24,83-86), formulas
(IoU tp/(tp+fp+fn), Dice 2tp/(2tp+fp+fn), Acc (tp+tn)/all with NO epsilon, :106-108) and macro mean over
non-ignored classes (:111-113).  The per-image argmax + one-hot + boolean reductions (:65-75) run as one
HIP kernel (ops.confusion_matrix: argmax takes the first maximum like torch.argmax) producing the exact
integer confusion matrix from which TP/FP/FN/TN follow."""
import torch

from . import ops


class MetricsHistory:
    def __init__(self, num_classes: int, ignore_index: int = None, device: str = 'cpu'):
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self.total_tp = torch.zeros(num_classes, dtype=torch.float64, device='cpu')
        self.total_fp = torch.zeros(num_classes, dtype=torch.float64, device='cpu')
        self.total_fn = torch.zeros(num_classes, dtype=torch.float64, device='cpu')
        self.total_tn = torch.zeros(num_classes, dtype=torch.float64, device='cpu')
        self.epoch_mean_dice_history = []
        self.epoch_mean_iou_history = []
        self.epoch_mean_acc_history = []
        self.epoch_per_class_dice_history = []
        self.epoch_per_class_iou_history = []
        self.epoch_per_class_acc_history = []
        self.last_per_class_iou = None
        self.last_per_class_dice = None
        self.last_per_class_acc = None
        self.mask = torch.ones(num_classes, dtype=torch.bool)
        self._dev_M = None              # device-resident confusion sums of accumulate_deferred (not pickled)
        self._dev_pixels = 0
        if self.ignore_index is not None and 0 <= self.ignore_index < self.num_classes:
            self.mask[self.ignore_index] = False

    def __getstate__(self):                 # checkpoints ("history": agg) carry the host totals only
        self.flush()
        d = dict(self.__dict__)
        d["_dev_M"], d["_dev_pixels"] = None, 0
        return d

    def reset(self):
        if self._dev_M is not None:
            self._dev_M.zero_()
        self._dev_pixels = 0
        self.total_tp.zero_()
        self.total_fp.zero_()
        self.total_fn.zero_()
        self.total_tn.zero_()

    @staticmethod
    def counts_from_confusion(M, num_pixels):
        """M[pred, label] (int64) -> tp, fp, fn, tn per class (float64, CPU)."""
        M = M.to(torch.float64).cpu()
        tp = M.diagonal().clone()
        fp = M.sum(dim=1) - tp          # predicted k, label != k
        fn = M.sum(dim=0) - tp          # label k, predicted != k
        tn = float(num_pixels) - tp - fp - fn
        return tp, fp, fn, tn

    def accumulate(self, pred: torch.Tensor, label: torch.Tensor):
        """pred: logits/probabilities (C,H,W) [or (1,C,H,W)]; label: (H,W) [or (1,H,W)] int64."""
        p = pred.squeeze(0) if pred.dim() == 4 else pred
        lab = label.squeeze(0) if label.dim() == 3 else label
        lab_l = lab.long()
        # F.one_hot in the reference (MetricsHistory.py:68) rejects labels outside [0, C): keep that contract
        if lab_l.numel() and (int(lab_l.min()) < 0 or int(lab_l.max()) >= self.num_classes):
            raise RuntimeError("Class values must be smaller than num_classes.")
        M = ops.confusion_matrix(p, lab_l, self.num_classes)
        tp, fp, fn, tn = self.counts_from_confusion(M, lab_l.numel())
        self.total_tp += tp
        self.total_fp += fp
        self.total_fn += fn
        self.total_tn += tn

    def accumulate_deferred(self, pred: torch.Tensor, label: torch.Tensor):
        """`accumulate` without a host round trip per image (the eval loops' device path): counts are added to a
        confusion matrix that stays on the GPU; `flush()` (called by compute_epoch_metrics) moves the sums over once.
        The label range check of `accumulate` is made there too: a label outside [0, C) leaves the matrix short of
        the pixel count."""
        if not pred.is_cuda:
            return self.accumulate(pred, label)
        p = pred.squeeze(0) if pred.dim() == 4 else pred
        lab = label.squeeze(0) if label.dim() == 3 else label
        dev = p.device
        if self._dev_M is None or self._dev_M.device != dev:
            self.flush()
            self._dev_M = torch.zeros((ops._lib.MAX_CLASSES, ops._lib.MAX_CLASSES), dtype=torch.int64, device=dev)
        ops.confusion_matrix(p, lab.to(dev), self.num_classes, out=self._dev_M)
        self._dev_pixels += lab.numel()

    def flush(self):
        if self._dev_M is None or self._dev_pixels == 0:
            return
        M = self._dev_M[:self.num_classes, :self.num_classes].cpu()
        pixels, self._dev_pixels = self._dev_pixels, 0
        self._dev_M.zero_()
        if int(M.sum()) != pixels:
            raise RuntimeError("Class values must be smaller than num_classes.")
        tp, fp, fn, tn = self.counts_from_confusion(M, pixels)
        self.total_tp += tp
        self.total_fp += fp
        self.total_fn += fn
        self.total_tn += tn

    def compute_epoch_metrics(self, epsilon: float = 1e-6):
        self.flush()
        tp, fp, fn, tn = self.total_tp, self.total_fp, self.total_fn, self.total_tn
        per_class_iou = tp / (tp + fp + fn)
        per_class_dice = (2 * tp) / (2 * tp + fp + fn)
        per_class_acc = (tp + tn) / (tp + tn + fp + fn)
        mean_iou = per_class_iou[self.mask].mean().item()
        mean_dice = per_class_dice[self.mask].mean().item()
        mean_acc = per_class_acc[self.mask].mean().item()
        self.epoch_mean_iou_history.append(mean_iou)
        self.epoch_mean_dice_history.append(mean_dice)
        self.epoch_mean_acc_history.append(mean_acc)
        self.epoch_per_class_iou_history.append(per_class_iou.numpy())
        self.epoch_per_class_dice_history.append(per_class_dice.numpy())
        self.epoch_per_class_acc_history.append(per_class_acc.numpy())
        self.last_per_class_iou = per_class_iou
        self.last_per_class_dice = per_class_dice
        self.last_per_class_acc = per_class_acc
        return mean_dice, mean_iou, mean_acc

    def to(self, device):
        self.total_tp = self.total_tp.to(device)
        self.total_fp = self.total_fp.to(device)
        self.total_fn = self.total_fn.to(device)
        self.total_tn = self.total_tn.to(device)
        self.mask = self.mask.to(device)
        if self.last_per_class_iou is not None:
            self.last_per_class_iou = self.last_per_class_iou.to(device)
        if self.last_per_class_dice is not None:
            self.last_per_class_dice = self.last_per_class_dice.to(device)
        if self.last_per_class_acc is not None:
            self.last_per_class_acc = self.last_per_class_acc.to(device)

    def get_ignore_index(self):
        return self.ignore_index

    def get_num_classes(self):
        return self.num_classes

    def get_mean_dice_history(self):
        return self.epoch_mean_dice_history

    def get_mean_iou_history(self):
        return self.epoch_mean_iou_history

    def get_mean_acc_history(self):
        return self.epoch_mean_acc_history

    def get_class_dice_history(self):
        return self.epoch_per_class_dice_history

    def get_class_iou_history(self):
        return self.epoch_per_class_iou_history

    def get_class_acc_history(self):
        return self.epoch_per_class_acc_history

    def get_last_per_class_dice(self):
        return self.last_per_class_dice

    def get_last_per_class_iou(self):
        return self.last_per_class_iou

    def get_last_per_class_acc(self):
        return self.last_per_class_acc
