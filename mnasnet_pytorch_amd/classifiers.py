"""Drop-in for the mnasnet branch of the reference's ``src/models/classifiers.py``:
``load_model`` (classifiers.py:7-17) and ``FineTuneModelPool`` (classifiers.py:19-111).

The feature extractor is the HIP engine (``Mnasnet(cut_channels_first=False).features``, exactly what
train.py:194-207 builds); the global average pool is fused into the engine's last kernel and the MLP head runs on
the HIP library too (head.py / csrc/mnas_head.hip).  ``pooling`` and ``classifier`` stay ordinary
``nn.Module`` attributes (state_dict keys, ``.parameters()``, replacing them all work); a head that is not a
Dropout/Linear/ReLU chain simply runs as the PyTorch module it is.  The resnet branches of the reference need
torchvision and are outside the hot path: they raise."""
import torch.nn as nn

from .head import NativeHead
from .mnasnet import Mnasnet


def load_model(arch="resnet18", pretrained=True):
    if arch.startswith("mnasnet"):
        model = Mnasnet(cut_channels_first=False)       # classifiers.py:13 (pretrained is ignored there too)
        print("Mnasnet initialized")
        return model
    raise ValueError("Finetuning not supported on this architecture yet")   # resnet*: torchvision, out of scope


class FineTuneModelPool(nn.Module):
    def __init__(self, original_model, arch, num_classes, classifier_config):
        super().__init__()
        self.num_classes = num_classes
        if not arch.startswith("mnasnet"):
            raise ValueError("Finetuning not supported on this architecture yet")
        self.features = original_model.features          # classifiers.py:47 (re-parented, same object)
        final_feature_map = 320                          # classifiers.py:48
        self.pooling = nn.Sequential(nn.AdaptiveAvgPool2d(1))
        self.modelName = "mnasnet"
        if classifier_config == "256":
            self.classifier = nn.Sequential(nn.Dropout(), nn.Linear(final_feature_map, 256), nn.ReLU(inplace=True),
                                            nn.Dropout(), nn.Linear(256, num_classes))
        elif classifier_config == "512_256":
            self.classifier = nn.Sequential(nn.Dropout(), nn.Linear(final_feature_map, 512), nn.ReLU(inplace=True),
                                            nn.Dropout(), nn.Linear(512, 256), nn.ReLU(inplace=True), nn.Dropout(),
                                            nn.Linear(256, num_classes))
        elif classifier_config == "320":
            self.classifier = nn.Sequential(nn.Dropout(p=0.2, inplace=True), nn.Linear(final_feature_map, num_classes))
        elif classifier_config == "512":
            self.classifier = nn.Sequential(nn.Dropout(), nn.Linear(final_feature_map, 512), nn.ReLU(inplace=True),
                                            nn.Dropout(), nn.Linear(512, num_classes))
        else:
            raise ValueError("Finetuning not supported on this architecture yet")
        self.input_norm_on_device = False     # see normalize_on_device()
        self.fuse_pool = True        # see forward()
        self.native_head = True      # classifier on csrc/mnas_head.hip when it is a Dropout/Linear/ReLU chain
        self._head = None
        self._head_key = None
        self.mean = (0.485, 0.456, 0.406)             # classifiers.py:91-92 (the reference stores them and never uses them:
        self.std = (0.229, 0.224, 0.225)              # its datasets normalise on the CPU)

    # The NativeHead cache holds the ctypes library handle: it is process-local state, not model state.  Keep it out of
    # pickles / deep copies (torch.save(model), copy.deepcopy(model) for EMA or best-model copies); it is rebuilt lazily.
    def __getstate__(self):
        st = self.__dict__.copy()
        st["_head"], st["_head_key"] = None, None
        return st

    def normalize_on_device(self, enable=True):
        """Move the dataset's ``transforms.Normalize(self.mean, self.std)`` (datasets.py:474-516) into the stem conv's input load:
        after this the model takes UN-normalised images -- float in [0, 1] or raw uint8 (then the batch crosses PCIe at a quarter
        of the bytes of train.py:427's fp32 upload).  Off by default: the reference's loaders hand over normalised floats."""
        self.input_norm_on_device = bool(enable)       # module state: survives pickling / deepcopy, re-applied to a rebuilt engine
        self._sync_input_norm()
        return self

    def _sync_input_norm(self):
        """Make the (lazily built, never pickled) engine of ``features`` agree with ``input_norm_on_device``."""
        if not hasattr(self.features, "_engine"):
            return
        want = (tuple(self.mean), tuple(self.std)) if getattr(self, "input_norm_on_device", False) else None
        eng = self.features._engine()
        if getattr(eng, "_in_norm_key", None) != want:
            if want is None:
                eng.set_input_normalization(None, None)
            else:
                eng.set_input_normalization(*want)
            eng._in_norm_key = want

    def freeze(self):
        print("Features frozen")
        for p in self.features.parameters():
            p.requires_grad = False

    def unfreeze(self):
        print("Features unfrozen")
        for p in self.features.parameters():
            p.requires_grad = True

    def _pool_is_global_average(self):
        if len(self.pooling) != 1 or not isinstance(self.pooling[0], nn.AdaptiveAvgPool2d):
            return False
        return self.pooling[0].output_size in (1, (1, 1))

    def forward(self, x):
        # classifiers.py:107-111.  features -> AdaptiveAvgPool2d(1) -> flatten is ONE engine call: the pool is fused with the
        # last BatchNorm+ReLU (mnas_pool_act) and the N x 320 x H/32 x W/32 fp32 feature map is never materialised; backward
        # starts from the pooled gradient (mnas_pool_bwd).  Any other pooling module takes the two-step path.
        self._sync_input_norm()
        hooked = bool(self.features._forward_hooks or self.features._forward_pre_hooks or self.pooling._forward_hooks
                      or self.pooling._forward_pre_hooks)       # hooks on features / pooling must fire: take the module path
        if self.fuse_pool and not hooked and hasattr(self.features, "_engine") and self._pool_is_global_average():
            f = self.features._engine().forward(x, pooled=True)
        else:
            f = self.pooling(self.features(x))
        f = f.view(f.size(0), -1)
        head = self._native_head() if (self.native_head and f.is_cuda) else None
        if head is not None:
            return head.apply(f)
        return self.classifier(f)

    def _native_head(self):
        """NativeHead over the CURRENT classifier modules (rebuilt if the Sequential or one of its children was replaced);
        None if the classifier is not a Dropout/Linear/ReLU chain."""
        c = self.classifier
        key = (id(c),) + tuple(id(m) for m in c.children()) + tuple((m.p if isinstance(m, nn.Dropout) else None) for m in c.children())
        if key != self._head_key:
            self._head, self._head_key = NativeHead.build(c), key
        return self._head
