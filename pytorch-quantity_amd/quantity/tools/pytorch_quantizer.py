"""Calibration orchestrator: discover the cared tensors of a model, run the two-pass activation
calibration (abs-max, then 2048-bin histograms, then the KL threshold sweep), write feat.table;
quantise weights, write weight.table and the per-parameter JSON files.

Drop-in for reference quantity/tools/pytorch_quantizer.py (Quantity :19, build_net_structure
:65-197, get_cared_op_names :199-204, prune_net_info :207-249, preprocess :252-284, net_forward
:288-296, get_merge_groups :298-341, activation_quantize :345-489, regist_hook_outfeature :491-524,
init_dir :529-548, rewrite_weight :553-590, weight_quantize :592-677, dilation_to_zero_padding
:679-693): same class, method names and call order, same config files (cwd-relative
../tools/configs.yml and ./user_configs.yml), same output files byte for byte.

MI355X design (what is different underneath):
  * hooked activations never leave HBM: the hooks keep device tensors, and each pass ends in ONE
    segmented HIP launch over all cared tensors of the batch (the reference copies every hooked
    tensor to the host and forks a process pool per batch);
  * graph edges come from tensor identity during one traced forward (value fingerprints, the
    reference's `tid`, are only the fallback), so all-positive inputs to ReLU etc. need no special case;
  * pass 2 re-uses activations of pass 1 wherever HBM allows: the bin width needs the global maximum
    first, so the reference runs every image through the network twice; an MI355X has 288 GB, so
    pass 1 keeps cared activations alive within a budget (FQ_ACT_CACHE_GB) -- either whole batches, or
    (usually better) the deepest suffix of EVERY batch, in which case the second forward stops at
    the last tensor that was not kept (_plan_cache).  Same tensors, same integers;
  * data-parallel calibration: when torch.distributed is initialised the calibration batches are
    dealt round-robin to the ranks (one process per GPU) and the per-tensor maxima / histograms
    are combined with one MAX and one SUM all-reduce (RCCL over xGMI); integer sums and maxima are
    order independent, so the tables are bit-identical for any number of GPUs.  Rank 0 writes files.
"""
import math
import os
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import yaml

from common.quantity import DistributionCollector, Quantizer, walk_dirs, merge_bn, tid  # noqa: F401
from common.quantity import _native, _float_conv
from .rewriter import BiasReWriter
from ._jsonio import dump_int_array

__all__ = ["Quantity"]


def _load_yaml(path):
    with open(path) as fh:
        return yaml.safe_load(fh)


def _dist_state():
    """(rank, world) of the default process group, (0, 1) when not distributed."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


from ._fused_forward import (_FusedForward, _StopForward, _flag, _set_flag, _RELU_VERIFIED, _POOL_VERIFIED, _POOL_OFF,  # noqa: E402,F401
                             _FUSION_VERIFIED)
from ._file_inputs import _FileInputs, _FileGroup, _dist_on  # noqa: E402,F401
from ._hook_state import _AFTER_FORWARD, _DeferralProbe, _EagerStats, _HookState  # noqa: E402,F401


class Quantity(_FusedForward, _FileInputs):

    # the statistics engine; tests substitute oracle-backed doubles to exercise the host logic on CPU
    collector_cls = DistributionCollector
    quantizer_cls = Quantizer
    channel_collector_cls = None    # None: common.quantity.channel_collector.ChannelCollector (activation_quantize_per_channel)
    profile_phases = False      # synchronise at phase boundaries so that .timings are device times
    # statistics kernels on a side stream under the next forward: measured on MI355X (ResNet-50, batch
    # 128) +0.8 % images/s, while the histogram kernel drops from 5.2 to 3.5 TB/s under contention and
    # deferred frees push the footprint from 106 to 150 GB -- not worth it, off by default.
    overlap_streams = False
    # Hooked activations are handed to the statistics kernels through _EagerStats: bytes gathered per launch.
    # None = decide from the first forward (which always runs one launch per tensor): everything in ONE launch after the
    # forward when no hooked tensor is modified in place afterwards, else one launch per tensor, from the hook, and no
    # activation cache.  Smaller groups (e.g. 96 << 20) were measured on ResNet-50: no gain.
    # Engines without `supports_partial` (the CPU test doubles) take the statistics after the forward.
    stats_group_bytes = None
    # Pass 1: a hooked nn.Conv2d with a bias runs as convolution-without-bias + fq_bias_add_absmax_f32 (the bias add torch
    # would launch anyway, with the running abs-max folded in), so its output is not read a second time for the maximum.
    # Every module is checked bit for bit against torch's own forward the first time it is used; one mismatch turns the
    # whole mechanism off for the run.
    fuse_bias_absmax = True
    # ... and when an out-of-place nn.ReLU consumes that output directly, the same kernel writes the ReLU's result too
    # (one more 4-byte write instead of the ReLU's own 8-byte pass); the patched ReLU.forward hands it out.
    fuse_relu = True
    # Pass 2: the same producers histogram their own output (fq_bias_add_hist_f32 / fq_add_hist_f32) for every tensor the
    # second forward re-computes, instead of handing it to the streaming histogram kernel right after writing it (which
    # costs a second 4 B/element read and runs that kernel against the write-back of its own input).
    # Only modules whose decomposition was verified in pass 1 take part.
    fuse_hist = True
    # The float 1x1 convolutions themselves (36 of ResNet-50's 53) and the 7x7 stride-2 stem: fq_conv1x1_f32 /
    # fq_conv_stem_f32 compute them on the fp32 matrix cores (exact fp32, fixed summation order) with the bias, the statistic
    # of the pass and the following ReLU in the epilogue, so for these layers there is no library convolution and no
    # bias-add pass at all.  Each module is checked once per process, on its first batch, against an independent
    # implementation of the same fp32 mathematics -- torch.matmul for the 1x1 layers, torch's Conv2d.forward for the stem
    # (|difference| <= 1e-5 * (|W|*|x| + |b|): summation order only; the abs-max and the ReLU copy bit for bit); a module
    # that disagrees keeps the path above.
    # nn.MaxPool2d and a global nn.AvgPool2d of the model run on fq_maxpool2d_f32 / fq_avgpool_global_f32 during a GPU
    # calibration: the same bits as torch (checked once per module, torch.equal), at 2-7x torch's rate.
    own_pools = os.environ.get("FQ_OWN_POOLS", "1") != "0"
    # Pass 1: the last 1x1 convolution of a residual block, the Eltwise that adds its output to the shortcut and the ReLU behind
    # it run as ONE kernel (fq_conv1x1_add_f32) -- the convolution's output and the sum are written to HBM only if pass 2's
    # cache wants them: 8 bytes per element instead of 20.  A convolution is deferred to its Eltwise only after the poison
    # probe (_DeferralProbe) has shown that nothing else reads its output.
    fuse_conv_add = os.environ.get("FQ_FUSE_CONV_ADD", "1") != "0"
    # A convolution whose output only an out-of-place nn.ReLU reads (same proof) writes that ReLU's result and not its own
    # output, unless pass 2's cache wants the tensor: 4 bytes per element instead of 8.
    skip_unread_outputs = os.environ.get("FQ_SKIP_UNREAD", "1") != "0"
    # True: every hooked tensor is written to HBM even when nothing of this calibration will read it again (for observers that
    # tape the hooked tensors: the oracle-replay tests)
    materialize_all = False
    own_conv1x1 = _float_conv.enabled()                              # FQ_OWN_CONV1X1=0: A/B against the library convolutions
    # A residual sum both of whose operands pass 2's cache keeps anyway (conv3's output; the shortcut = the previous block's ReLU
    # output or the projection) is not written by pass 1: the cache keeps the shortcut in the sum's place and pass 2 histograms
    # (conv3 output, conv3 output + shortcut) in one pass over the pair (fq_hist2048_pair_seg).  Same integers.
    pair_hist = os.environ.get("FQ_PAIR_HIST", "1") != "0"
    # ... and a shortcut that is itself the ReLU output of an earlier such sum (every identity block) is not kept either: pass 2
    # walks the stage's chain in registers (fq_hist2048_chain_seg: S_k = y_k + relu(S_(k-1))) -- the cache then holds conv3's outputs
    # and one shortcut per stage, and what it no longer spends on sums it spends on earlier layers (a shorter second forward).
    pair_chain = os.environ.get("FQ_PAIR_CHAIN", "1") != "0"

    def __init__(self, model):
        assert os.path.isfile("../tools/configs.yml"), "./configs.yml"
        assert os.path.isfile("./user_configs.yml"), "../test/user_configs.yml"
        self.config = _load_yaml("../tools/configs.yml")
        self.user_config = _load_yaml("./user_configs.yml")
        self.init_dir()

        settings = self.config["SETTINGS"]
        self.device = self.user_config["SETTINGS"]["DEVICE"]
        if self.device == "gpu" and torch.cuda.is_available():
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                # under torchrun each rank already selected its own GPU
                torch.cuda.set_device(self.user_config["SETTINGS"]["GPU"])
        self._cared_op_type = settings["CARE_OP_TYPE"]
        self._all_op_type = settings["ALL_OP_TYPE"]
        self._allow_same_tid_op_type = settings["ALLOW_SAME_TID_OP_TYPE"]
        self._merge_op_type = settings["MERGE_OP_YTPE"]
        self._max_img_num = settings["MAX_CALI_IMG_NUM"]
        print("max_img_num", self._max_img_num)
        self.model = model
        self.input_size = tuple(int(v) for v in self.user_config["MODEL"]["INPUT_SHAPE"].split(","))
        self.layers_num = 0
        self.name_to_param = OrderedDict()
        self.net_info = self.build_net_structure(self.model, self.input_size, self.device)
        self.cared_op_layer_names = self.get_cared_op_names(self.model)
        self._DKL_weight = False
        self.timings = {}

    # ------------------------------------------------------------------------------------------
    # graph discovery
    # ------------------------------------------------------------------------------------------
    def _model_device(self, model):
        for p in model.parameters():
            return p.device
        return torch.device("cpu")

    def build_net_structure(self, model, input_size, device="cpu"):
        """One traced forward on random input.  Every module whose type is in ALL_OP_TYPE becomes a
        node "<ClassName>_<ordinal>" (ordinal = 1-based execution order); its inputs are the nodes
        that produced its input tensors.  Nodes not in CARE_OP_TYPE are then pruned, rewiring
        through single-input chains.  Returns OrderedDict name -> {'inputs': [...], 'type': str}.
        """
        assert device.lower() in ("gpu", "cpu"), "Input device is not valid, please specify 'gpu' or 'cpu'"
        dev = self._model_device(model)
        shapes = [input_size] if isinstance(input_size, tuple) else list(input_size)
        x = [torch.rand(*s, device=dev) for s in shapes]
        trace = []            # (name, type, [input tensors], output tensor), execution order
        handles = []

        def on_forward(module, inputs, output):
            kind = type(module).__name__
            trace.append(("%s_%i" % (kind, len(trace) + 1), kind, [t for t in inputs if torch.is_tensor(t)], output))

        for m in model.modules():
            if type(m).__name__ in self._all_op_type:
                handles.append(m.register_forward_hook(on_forward))
        # (only which tensor OBJECT reaches which module is taken from this forward: on the GPU its convolutions run on the own
        #  kernels, unchecked, so that a fresh process does not pay the convolution library's first-use search for a trace.
        #  Shape-only "meta" stand-ins would launch nothing at all, but their first use costs a fresh process more than this
        #  forward does -- 0.47 s against 0.22 s -- and the libraries' start-up then lands in the first calibration.)
        try:
            with torch.no_grad(), _float_conv.own_convs(model):
                model(*x)
        finally:
            for h in handles:
                h.remove()
        self.layers_num = len(trace)

        producer = {}          # id(tensor) -> node name; tensors are kept alive by `trace`
        fingerprint = {}       # tid -> node name, fallback when a tensor object was re-wrapped; filled when first asked for
        fingerprinted = [0]

        def by_fingerprint(t, upto):
            for j in range(fingerprinted[0], upto):            # the outputs of the nodes before this one, first producer wins
                o = trace[j][3]
                if torch.is_tensor(o) and o.numel():
                    fingerprint.setdefault(tid(o), trace[j][0])
            fingerprinted[0] = max(fingerprinted[0], upto)
            return fingerprint.get(tid(t))

        net = OrderedDict()
        for i, (name, kind, ins, out) in enumerate(trace):
            inputs = []
            for t in ins:
                src = producer.get(id(t))
                if src is None:
                    src = by_fingerprint(t, i) if (i and t.numel()) else None     # (nothing was produced before node 0)
                if src is not None:
                    inputs.append(src)
                elif i != 0:
                    raise AssertionError("Can't find the input tensor of {} \n {}".format(name, net))
            if torch.is_tensor(out):
                if id(out) in producer and kind not in self._allow_same_tid_op_type:
                    if any(id(out) == id(t) for t in ins):
                        raise ValueError("Same input and output id, the op {} is useful?".format(name))
                    raise AssertionError("Some layers returned same tensor.")
                producer[id(out)] = name           # a pass-through op becomes the newest producer
            net[name] = {"inputs": inputs, "type": kind}
        keep = [n for n, info in net.items() if info["type"] in self._cared_op_type]
        return self.prune_net_info(net, keep)

    def get_cared_op_names(self, model):
        return [name for name, module in model.named_modules() if type(module).__name__ in self._cared_op_type]

    def prune_net_info(self, net_info, keep_node_list):
        """Drop the nodes that are not kept; an edge into a dropped node is re-pointed at the
        nearest kept ancestor (dropped nodes must be single-input)."""
        dropped = set(net_info.keys()) - set(keep_node_list)

        def nearest_kept(name):
            ins = net_info[name]["inputs"]
            assert len(ins) <= 1, (ins, name)
            if not ins:
                return None
            return nearest_kept(ins[0]) if ins[0] in dropped else ins[0]

        pruned = OrderedDict()
        for name, info in net_info.items():
            if name in dropped:
                continue
            inputs = [nearest_kept(s) if s in dropped else s for s in info["inputs"]]
            pruned[name] = {"inputs": inputs, "type": info["type"]}
        return pruned

    # ------------------------------------------------------------------------------------------
    # input side
    # ------------------------------------------------------------------------------------------
    def preprocess(self, image):
        """PRE_PROCESS.IMG: 1 = the item is a (data, label) pair from a loader; 2 = path of a .npy
        holding one CHW image; 0 = path of an image file.

        Mode 0 in the reference (pytorch_quantizer.py:259-271) cannot run as shipped (`slef`,
        `image_path` NameErrors), so there is nothing to be bit-compatible with; this follows what it
        spells out: cv2.imread (BGR, uint8) -> cv2.resize(RESIZE) -> float32 - MEAN -> CHW -> [1,3,H,W]
        (SCALE is read and never applied there, and is not applied here).  Decoding uses Pillow;
        the resize is half-pixel bilinear rounded back to uint8, OpenCV's INTER_LINEAR convention."""
        mode = int(self.user_config["PRE_PROCESS"]["IMG"])
        if isinstance(image, _FileGroup):
            return self._preprocess_group(image, mode)
        if mode == 1:
            img, _ = image
            return img
        if mode == 2:
            arr = torch.as_tensor(np.load(image))
            return arr.view(1, *arr.shape)
        if mode == 0:
            from PIL import Image
            cfg = self.user_config["PRE_PROCESS"]["IMG_SET"]
            mean = float(cfg["MEAN"])
            width, height = (int(float(v)) for v in str(cfg["RESIZE"]).split(","))
            try:
                rgb = np.asarray(Image.open(image).convert("RGB"))
            except (OSError, ValueError):
                return False                                   # the reference returns False for unreadable files
            bgr = torch.from_numpy(np.ascontiguousarray(rgb[:, :, ::-1])).permute(2, 0, 1)[None].float()
            if bgr.shape[-2:] != (height, width):
                bgr = torch.nn.functional.interpolate(bgr, size=(height, width), mode="bilinear", align_corners=False)
                bgr = torch.clamp(torch.round(bgr), 0, 255)
            return bgr - mean
        print("input option set wrong:", mode)
        return None

    def net_forward(self, net, image_path):
        img = image_path if torch.is_tensor(image_path) else self.preprocess(image_path)
        if self.device == "gpu" and img.device.type != "cuda":
            img = img.cuda(non_blocking=True)
        with torch.no_grad():
            try:
                net(img)
            except _StopForward:
                pass

    # ------------------------------------------------------------------------------------------
    # merge groups
    # ------------------------------------------------------------------------------------------
    def get_merge_groups(self, net):
        """For every Eltwise / Concat node, deepest first: the cared tensors feeding it."""
        merge_layers = [n for n, info in net.items() if info["type"] in self._merge_op_type]
        merge_layers.reverse()
        print("merge layers:", merge_layers)
        seen = set()

        def cared_bottoms(name):
            if name in seen:
                return []
            seen.add(name)
            found = []
            for b in net[name]["inputs"]:
                if net[b]["type"] in self._cared_op_type:
                    found.append(b)
                else:
                    found.extend(cared_bottoms(b))
            return found

        groups = []
        for layer in merge_layers:
            bottoms = cared_bottoms(layer)
            print(layer, bottoms)
            if bottoms:
                groups.append(bottoms)
        return groups

    @staticmethod
    def reserve_pool(fraction=0.80, device=None, keep_free=8 << 30):
        """EXTENSION (no reference counterpart): grow the caching allocator's pool of `device` to `fraction` of its HBM, once, OUTSIDE
        any calibration -- the service-mode switch.  A calibration never grows its pool for the activation cache (28 ms per GB of
        fresh hipMalloc costs more than a cached GB saves, see _activation_cache_budget), so a fresh process runs pass 2 on second
        forwards; a process that calibrates repeatedly (or one big job) calls this first and every later
        activation_quantize() keeps pass 1's activations in the pool instead.  Returns the bytes the pool holds afterwards
        (torch.cuda.memory_reserved()); a no-op on a host without a GPU.  Never takes the last `keep_free` bytes of the device."""
        if not torch.cuda.is_available():
            return 0
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        free_b, total_b = torch.cuda.mem_get_info(dev)
        pooled_b = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        # one block of everything still missing: the allocator keeps it whole and carves the cache's tensors out of it
        grow = min(int(total_b * float(fraction)) - torch.cuda.memory_reserved(dev), free_b - int(keep_free))
        if grow > (64 << 20):
            block = torch.empty(grow, dtype=torch.uint8, device=dev)
            del block
        return int(torch.cuda.memory_reserved(dev))

    def _activation_cache_budget(self):
        """Bytes of HBM that pass 1 may keep alive for pass 2 (0 disables the cache)."""
        if self.device != "gpu" or not torch.cuda.is_available():
            return 0
        env = os.environ.get("FQ_ACT_CACHE_GB")
        if env is not None:
            return int(float(env) * (1 << 30))
        free, total = torch.cuda.mem_get_info()
        reserved, live = torch.cuda.memory_reserved(), torch.cuda.memory_allocated()
        pooled = reserved - live                   # held by the caching allocator, reusable by us
        # What a cache costs a COLD process is growing the caching allocator: 28 ms per GB of fresh hipMalloc on MI355X /
        # ROCm 7.2 (bench.py's fresh-process run: pass 1 of 5 120 ResNet-50 images 4.38 s with an 86 GB cache against
        # 0.67 s in a warm pool; scripts/alloc_probe.py: 10-30 ms per GB).  What a cached GB saves is at most ~8 ms (the
        # deepest activations: most forward time per byte; whole batches: 2 ms per GB).  So a process never GROWS its pool
        # for the cache (round 1 allowed itself 96 GB and ran the 5 120-image config at 1 040 images/s cold).
        # Memory the allocator already holds (a long-running calibration service, or bench.py after its warm-up) costs
        # nothing to use: all of it except 1/16 of the device, which stays with the forward's own transient tensors (the
        # part of HBM the pool does not cover absorbs anything beyond that).
        warm = pooled - (total >> 4)
        return max(warm, 0)

    def _cache_entry(self, named_feats, keep):
        """What pass 1 keeps of the forward that just ran: the hooked tensors named in `keep` (None: all) -- with every sum that
        was not written replaced by its pair, ("pair", sum key) -> (key of conv3's output, that tensor, the shortcut tensor or None,
        the shortcut's version, whether conv3's output is itself among the kept tensors, src).  A shortcut that is the ReLU output
        of an earlier sum of this very entry (src = that sum's key) is not kept at all: pass 2 walks the chain in registers
        (fq_hist2048_chain_seg), so a stage of identity blocks holds one shortcut, its head's.
        Returns (entry, bytes it holds on to)."""
        pairs = self._hook_ctl.pairs
        entry, held, nbytes, chain_len, continued = {}, set(), 0, {}, set()

        def hold(t):                                       # (by address: the hooks keep detach() aliases of the model's tensors)
            if t.data_ptr() not in held:
                held.add(t.data_ptr())
                return t.numel() * t.element_size()
            return 0
        for n, t in named_feats.items():                   # (forward order: a sum's src comes before it)
            if keep is not None and n not in keep:
                continue
            p = pairs.get(n)
            if p is None:
                entry[n] = t
                nbytes += hold(t)
                continue
            conv_key, other, version, src = p
            # (the forward that ran before the plan existed may have left a sum to its pair whose conv3 output the plan does not
            #  keep: that one batch then holds the tensor privately, and pass 2 counts it where the prefix forward re-makes it)
            y_kept = keep is None or conv_key in keep
            # (a chain is linear: when one sum's ReLU output is the shortcut of two later blocks, the first of them continues the
            #  chain and the second keeps its shortcut as a tensor)
            chained = (src is not None and self.pair_chain and ("pair", src) in entry and chain_len.get(src, 0) < _native.CHAIN_MAX
                       and src not in continued
                       and other.shape == named_feats[conv_key].shape == entry[("pair", src)][1].shape)
            chain_len[n] = chain_len[src] + 1 if chained else 1
            if chained:
                continued.add(src)
            entry[("pair", n)] = (conv_key, named_feats[conv_key], None if chained else other, version, y_kept, src if chained else None)
            nbytes += hold(named_feats[conv_key]) + (0 if chained else hold(other))
        return entry, nbytes

    def _add_with_pairs(self, collector):
        """collector.add_to_distributions for dicts that may hold pairs (see _cache_entry): the pairs go through
        add_pairs_to_distributions, which also counts conv3's output -- that tensor then leaves the plain list; pairs chained on
        each other's ReLU output (a stage of identity blocks) go through add_chains_to_distributions, one launch per chain."""
        def add(feats):
            pairs = [(k[1], v) for k, v in feats.items() if isinstance(k, tuple)]
            if not pairs:
                return collector.add_to_distributions(feats)
            plain = dict((k, v) for k, v in feats.items() if not isinstance(k, tuple))
            chains, chain_of = [], {}                                         # [(head, [(y, row of y, row of sum), ...])]
            for sum_key, v in pairs:                                          # (forward order: a sum's src comes before it)
                conv_key, y, other, version, y_kept, src = v
                if y_kept and plain.pop(conv_key, None) is None:
                    raise RuntimeError("pass 2 holds the pair of %s without %s" % (sum_key, conv_key))
                block = (y, conv_key if y_kept else None, sum_key)
                if src is None:
                    if other._version != version:
                        raise RuntimeError("the shortcut of %s was written to after the add that pass 1 left to pass 2; set "
                                           "Quantity.pair_hist = False (FQ_PAIR_HIST=0)" % sum_key)
                    chain_of[sum_key] = len(chains)
                    chains.append((other, [block]))
                else:
                    if src not in chain_of or chains[chain_of[src]][1][-1][2] != src:
                        raise RuntimeError("pass 2 holds the pair of %s without the pair of %s it is chained to" % (sum_key, src))
                    chain_of[sum_key] = chain_of[src]
                    chains[chain_of[src]][1].append(block)
            if plain:
                collector.add_to_distributions(plain)
            # single blocks go out together in one launch, a stage's chain in one launch of its own
            collector.add_pairs_to_distributions([(b[0][0], head, b[0][1], b[0][2]) for head, b in chains if len(b) == 1])
            collector.add_chains_to_distributions([c for c in chains if len(c[1]) > 1])
        add.__name__ = "add_to_distributions"
        add.__self__ = collector
        return add

    def _forward_with_stats(self, item, fn, named_feats, extra=None):
        """One forward (possibly ended early by the cache plan) with fn applied to every hooked tensor it produced and
        to `extra` (tensors of the same batch kept from pass 1).  Returns the _EagerStats of that forward, or None
        when the engine takes its statistics after the forward."""
        limit = self._stats_limit
        if limit is None:
            self.net_forward(self.model, item)
            feats = dict(named_feats)
            feats.update(extra or {})
            self._on_stat_stream(fn, feats)
            return None
        eager = _EagerStats(fn, limit)
        eager.retain = self._hook_ctl.keep_feats
        self._hook_ctl.eager = eager
        self._hook_ctl.pairs, self._hook_ctl.sum_relu = {}, {}
        try:
            self.net_forward(self.model, item)
        finally:
            self._hook_ctl.eager = None
            waiting, self._hook_ctl.deferred = self._hook_ctl.deferred, {}
        if waiting:
            # the forward took a path the probe did not see: whatever read these tensors read memory nobody had written
            self._hook_ctl.defer_ok = {}
            raise RuntimeError("a convolution deferred to its Eltwise (Quantity.fuse_conv_add) was never consumed by it: this "
                               "model's control flow depends on its data; set Quantity.fuse_conv_add = False")
        eager.flush(extra)
        return eager

    def _stat_stream(self):
        """Side HIP stream for the abs-max / histogram launches, or None (CPU, or overlap disabled).
        The statistics kernels stream 4 B/element from HBM and use almost no ALU; the next batch's
        convolutions are compute bound -- run together they hide each other."""
        if not self.overlap_streams or self.device != "gpu" or not torch.cuda.is_available():
            return None
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream()
        return self._side_stream

    def _on_stat_stream(self, fn, feats):
        """Run fn(feats) on the side stream after everything queued so far on the current stream."""
        side = self._stat_stream()
        if side is None:
            fn(feats)
            return
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fn(feats)
        for v in feats.values():
            for t in (v if isinstance(v, tuple) else (v,)):          # (a pair entry holds its tensors in a tuple)
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(side)      # the allocator must not recycle it while the side stream reads

    def _join_stat_stream(self):
        side = getattr(self, "_side_stream", None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)

    @staticmethod
    def _ordinal(name):
        return 0 if name == "image" else int(name.rsplit("_", 1)[1])

    def _plan_cache(self, budget, n_owned, feats, cum_ms):
        """Decide what pass 1 keeps for pass 2 inside `budget` bytes of HBM.

        plan A  keep ALL cared activations of the first m batches; the other batches are recomputed in full.
        plan B  keep, for EVERY batch, the deepest suffix of cared activations that fits; pass 2 re-runs
                only the prefix of the network up to the last tensor that was not kept (the hook ends the
                forward there).  Early layers hold most of the bytes, late layers most of the depth, so
                a small cache removes a large part of the second forward for every image.
        The cheaper one by the measured time stamps of the first forward wins.  Either way pass 2
        histograms exactly the tensors pass 1 took the maxima of, or fresh ones from the same weights."""
        sizes = [(n, self._ordinal(n), t.numel() * t.element_size()) for n, t in feats.items()]
        per_batch_all = sum(b for _n, _o, b in sizes)
        t_full = max(cum_ms.values()) if cum_ms else 1.0
        m = min(n_owned if n_owned is not None else 1 << 30, budget // max(per_batch_all, 1))
        plan = {"kind": "A", "whole_batches": int(m), "keep": None, "stop_after": None}
        if n_owned is None or n_owned == 0 or not cum_ms:
            return plan
        cost_a = (n_owned - min(m, n_owned)) * t_full
        room = budget // n_owned
        keep, used = [], 0
        # What keeping a tensor costs.  A sum that pass 1 leaves to its pair (Quantity.pair_hist) costs its SHORTCUT instead -- nothing
        # when that is a hooked tensor kept in its own right (a projection's output); and when it is the ReLU output of an earlier
        # such sum (pair_chain) the cost is given back as soon as that earlier block's conv3 output joins the suffix, because
        # pass 2 then re-makes the shortcut instead of reading a kept one.
        pair_info = dict(self._hook_ctl.pairs) if self.pair_hist else {}
        hooked = set(t.data_ptr() for t in feats.values())       # (by address: feats holds detach() aliases of the model's tensors)
        refund = {}                                              # conv3 key of the src pair -> bytes given back when it is kept
        depth = {}

        def depth_of(n):                                         # position of a sum in its stage's chain (the kernel takes 6 blocks)
            if n not in depth:
                src = pair_info[n][3]
                depth[n] = depth_of(src) + 1 if src is not None and src in pair_info else 0
            return depth[n]
        for n, o, b in sorted(sizes, key=lambda e: -e[1]):       # deepest first
            cost = b
            p = pair_info.get(n)
            if p is not None:
                conv_key, other, _version, src = p
                cost = 0 if other.data_ptr() in hooked else other.numel() * other.element_size()
                if cost and src is not None and self.pair_chain and src in pair_info and depth_of(n) % _native.CHAIN_MAX:
                    refund[pair_info[src][0]] = refund.get(pair_info[src][0], 0) + cost
            if n == "image" or used + cost > room:
                break
            keep.append(n)
            used += cost - refund.pop(n, 0)
        if not keep:
            return plan
        early = [o for n, o, _b in sizes if n not in keep and n != "image"]
        stop_after = max(early) if early else 0
        cost_b = n_owned * (cum_ms.get(stop_after, 0.0) if stop_after else 0.0)
        forced = os.environ.get("FQ_CACHE_PLAN", "")           # "A" / "B": testing and A/B timing
        if (cost_b < cost_a and forced != "A") or forced == "B":
            plan = {"kind": "B", "whole_batches": 0, "keep": set(keep), "stop_after": stop_after,
                    "prefix_fraction": round((cum_ms.get(stop_after, 0.0) if stop_after else 0.0) / t_full, 3)}
        return plan

    def _sync(self):
        if self.device == "gpu" and torch.cuda.is_available():
            torch.cuda.synchronize()

    def _group_has_eltwise(self, group):
        return any(self.net_info[n]["type"] == "Eltwise" for n in group)

    # ------------------------------------------------------------------------------------------
    # activation calibration
    # ------------------------------------------------------------------------------------------
    def _calibration_items(self, images_files):
        """Yield the calibration items this rank owns.  Batches 0..MAX_CALI_IMG_NUM are used
        (i > MAX breaks: N+1 batches, pytorch_quantizer.py:381); with W ranks, batch i goes to
        rank i % W.  Sequences are indexed so that other ranks' batches are never materialised."""
        if getattr(images_files, "_fq_indexed", False):            # already (index, item) pairs of this rank
            for pair in images_files:
                yield pair
            return
        if not self._file_batching():
            for pair in self._owned_items(images_files):
                yield pair
            return
        # file inputs: file_batch consecutive files of this rank form one item, named by its first file's index
        group, first = _FileGroup(), None
        for i, item in self._owned_items(images_files):
            if not group:
                first = i
            group.append(item)
            if len(group) == self.file_batch:
                yield first, group
                group = _FileGroup()
        if group:
            yield first, group

    def _owned_items(self, images_files):
        rank, world = _dist_state()
        last = self._max_img_num
        if hasattr(images_files, "__getitem__") and hasattr(images_files, "__len__"):
            for i in range(rank, min(len(images_files), last + 1), world):
                yield i, images_files[i]
            return
        for i, item in enumerate(images_files):
            if i > last:
                break
            if i % world == rank:
                yield i, item

    def _skip(self, images_files, skip_ids):
        """The calibration set minus the batches whose activations are already cached (never fetched)."""
        outer = self

        class _View(object):
            def __iter__(self_inner):
                return ((i, it) for i, it in outer._calibration_items(images_files) if i not in skip_ids)

        view = _View()
        view._fq_indexed = True
        return view

    def activation_quantize(self, images_files):
        settings = self.config["SETTINGS"]
        table_file = self.config["OUTPUT"]["FEAT_BIT_TABLE"]
        rank, world = _dist_state()

        merge_groups = self.get_merge_groups(self.net_info)
        top_feat_names = ["image"] + list(self.net_info.keys())
        collector = self.collector_cls(top_feat_names, interval_num=settings["INTERVAL_NUM"],
                                       statistic=settings["STATISTIC"], worker_num=settings["WORKER_NUM"],
                                       debug=False)
        quantizer = self.quantizer_cls(top_feat_names, worker_num=settings["WORKER_NUM"], debug=False)
        named_feats, hooks = self.regist_hook_outfeature(self.model)
        self._collector, self._quantizer = collector, quantizer
        self._file_kept, self._file_kept_bytes = {}, 0
        patched = self._patch_fused_convs(self.model) if getattr(collector, "supports_partial", False) else []
        try:
            return self._calibrate(images_files, collector, quantizer, named_feats, merge_groups, top_feat_names,
                                   table_file)
        finally:
            # (an exception in pass 2 -- a data-loader error, an FqError -- must not leave the pass's modes set on the controller)
            ctl = self._hook_ctl
            ctl.fuse_collector, ctl.fuse_stat, ctl.own_plain, ctl.stop_after, ctl.eager = None, "max", False, None, None
            ctl.deferred = {}
            for m in patched:
                del m.forward
            for h in hooks:                 # (the reference never removes its hooks)
                h.remove()
            named_feats.clear()
            self._file_kept, self._file_kept_bytes = {}, 0
            self.__dict__.pop("_pinned_ring", None)

    def _calibrate(self, images_files, collector, quantizer, named_feats, merge_groups, top_feat_names, table_file):
        rank, world = _dist_state()
        t0 = time.perf_counter()

        # pass 1: running abs-max of every cared tensor; keep activations for pass 2 while HBM allows
        budget = self._activation_cache_budget()
        eager_ok = (self.device == "gpu" and torch.cuda.is_available() and not self.overlap_streams
                    and getattr(collector, "supports_partial", False))
        self._stats_limit = None if not eager_ok else 0
        n_owned = None
        if hasattr(images_files, "__len__") and hasattr(images_files, "__getitem__"):
            n_owned = len(range(rank, min(len(images_files), self._max_img_num + 1), world))
            if self._file_batching():
                n_owned = (n_owned + self.file_batch - 1) // self.file_batch
        ctl = self._hook_ctl
        plan = None
        cached, cached_ids, used = {}, set(), 0
        step_ms = []
        inplace = None                          # does a later module overwrite a hooked tensor?
        if eager_ok and self._training_state_modules():
            # a model in training mode (BatchNorm statistics, dropout): nothing but calibration data may pass through it, so
            # no probe -- one launch per tensor from inside the hooks (always correct), nothing cached, pass 2 not fused
            budget = 0
        elif eager_ok:
            # One probe forward on a random input of INPUT_SHAPE (what build_net_structure traces with; nothing is taken
            # from it) answers that BEFORE the first calibration batch: the first batch then already runs in its final
            # mode -- in particular, when nothing will be cached, without the hooks keeping a batch's 17 GB of tensors
            # alive (a fresh process would grow its allocator pool for them: up to 0.6 s of hipMalloc).
            # (its values are not used, so the convolutions the own kernels take run on them here too, unchecked: a
            # calibration then never enters the convolution library, whose first-use solver search is most of what a
            # fresh process used to wait for; every module is still checked on the first real batch)
            deferral = (_DeferralProbe() if (self.fuse_conv_add or self.skip_unread_outputs) and self.fuse_bias_absmax and self.fuse_relu
                        and self.own_conv1x1 and self._stat_stream() is None else None)
            inplace = self._probe_forward("unchecked", deferral)
            if deferral is not None and not inplace and (deferral.pairs or deferral.conv_out):
                self._probe_feats = named_feats
                try:
                    ctl.defer_ok, ctl.relu_only_ok = self._prove_deferral(deferral, dict(named_feats), self._probe_out)
                finally:
                    self._probe_feats = None
            self._probe_out = None
            named_feats.clear()
            if inplace:
                budget = 0                      # kept tensors would hold overwritten values: no cache, per-tensor launches
                self._stats_limit = 0
            else:
                self._stats_limit = _AFTER_FORWARD if self.stats_group_bytes is None else int(self.stats_group_bytes)
                if not budget:
                    ctl.keep_feats = False   # nothing will be cached: producers' tensors need not outlive their hook
        ctl.fuse_collector = collector if eager_ok and self.fuse_bias_absmax else None
        for i, item in self._device_items(images_files):
            ts = time.perf_counter()
            if budget and plan is None and self.device == "gpu" and torch.cuda.is_available():
                start = torch.cuda.Event(enable_timing=True)
                start.record()
                ctl.events = []
            # what pass 2 will want kept of THIS forward (a deferred convolution writes its output and the sum only then)
            if not budget:
                ctl.keep_names = set()
            elif plan is None:
                ctl.keep_names = None
            elif plan["kind"] == "A":
                ctl.keep_names = None if len(cached) < plan["whole_batches"] else set()
            else:
                ctl.keep_names = plan["keep"]
            self._forward_with_stats(item, collector.refresh_max_val, named_feats)
            if os.environ.get("FQ_DEBUG_STEP_TIMES"):
                self._sync()
                step_ms.append(round((time.perf_counter() - ts) * 1e3, 2))
            if not budget:
                continue
            if plan is None:
                cum_ms = {}
                if ctl.events:
                    self._sync()
                    cum_ms = {o: start.elapsed_time(ev) for o, ev in ctl.events}
                ctl.events = None
                plan = self._plan_cache(budget, n_owned, named_feats, cum_ms)
            need_all = sum(t.numel() * t.element_size() for t in named_feats.values())
            if plan["kind"] == "A":
                if len(cached) < plan["whole_batches"] and used + need_all <= budget:
                    cached[i], nbytes = self._cache_entry(named_feats, None)
                    cached_ids.add(i)
                    used += nbytes
            else:
                cached[i], nbytes = self._cache_entry(named_feats, plan["keep"])
                used += nbytes
        ctl.fuse_collector = None
        self._join_stat_stream()
        if _dist_on():                      # also at world size 1: same code path, trivial cost
            collector.all_reduce_max()
        distribution_intervals = collector.distribution_intervals      # (device -> host sync)
        t1 = time.perf_counter()

        # tensors that are added / concatenated must share one scale: the group's largest interval
        # (groups fed by an Eltwise are tied after the KL search instead)
        for group in merge_groups:
            assert len(group) > 1
            if self._group_has_eltwise(group):
                continue
            widest = 0
            for name in group:
                widest = max(widest, distribution_intervals[name])
            for name in group:
                distribution_intervals[name] = widest

        # pass 2: histograms with the final intervals
        print("Collect histograms of activations:")
        if (eager_ok and self.fuse_bias_absmax and self.fuse_hist and not ctl.fuse_off and ctl.fuse_verified
                and inplace is False and hasattr(collector, "prepare_distributions") and getattr(collector, "fused_hist_ok", True)):
            collector.prepare_distributions()
            ctl.fuse_collector, ctl.fuse_stat = collector, "hist"
        elif eager_ok and self.fuse_bias_absmax and self.own_conv1x1 and not ctl.fuse_off and ctl.fuse_verified:
            # Pass 2 without the producers' histograms (a later module overwrites hooked tensors in place, or fuse_hist is off):
            # the convolutions stay on the own kernels all the same (their plain form; the statistics come from the hooks), so that
            # pass 2 histograms the very values pass 1 took the maxima of.  They used to fall back to the convolution library here,
            # whose kernels do not give the same bits from call to call: the histograms of such a model differed by a handful of
            # elements from one calibration to the next (scripts/model_fuzz.py, round 5).
            ctl.own_plain = True
        add = self._add_with_pairs(collector) if ctl.pair_sums else collector.add_to_distributions
        if plan is not None and plan["kind"] == "B":
            ctl.stop_after = plan["stop_after"] if plan["stop_after"] else None
            try:
                for i, item in self._device_items(images_files):
                    if plan["stop_after"]:
                        # ends at the last tensor that was not kept; fresh and kept tensors go out in one launch
                        self._forward_with_stats(item, add, named_feats, extra=cached.pop(i))
                        continue
                    feats = {"image": self.preprocess(item) if not torch.is_tensor(item) else item}
                    if self.device == "gpu" and feats["image"].device.type != "cuda":
                        feats["image"] = feats["image"].cuda()
                    feats.update(cached.pop(i))
                    self._on_stat_stream(add, feats)
            finally:
                ctl.stop_after = None
        else:
            for i in sorted(cached):
                self._on_stat_stream(add, cached[i])
            for i, item in self._device_items(self._skip(images_files, cached_ids)):
                self._forward_with_stats(item, add, named_feats)
        self._join_stat_stream()
        del cached
        ctl.fuse_collector, ctl.fuse_stat, ctl.own_plain = None, "max", False
        if _dist_on():
            collector.all_reduce_hist()
        if self.profile_phases:
            self._sync()
        t2 = time.perf_counter()

        # a merged group is searched on the sum of its members' histograms
        pooled = [g for g in merge_groups if not self._group_has_eltwise(g)]
        distributions = collector.merged_distributions(pooled)
        quantizer.quantize(distributions, distribution_intervals)
        bits = quantizer.bits

        # eltwise_k = eltwise_{k-1} + conv: the conv takes the earlier eltwise's bit
        for group in merge_groups:
            assert len(group) > 1
            elt_idx = None
            for i, name in enumerate(group):
                if self.net_info[name]["type"] == "Eltwise":
                    elt_idx = i
            if elt_idx is not None:
                conv_idx = 1 - elt_idx
                print("bit conv:eltwise ", bits[group[conv_idx]], bits[group[elt_idx]])
                bits[group[conv_idx]] = bits[group[elt_idx]]
        t3 = time.perf_counter()

        lines = []
        first_op = True
        for i, feat_name in enumerate(top_feat_names):
            if feat_name == "image":
                line = "image " + str(bits["image"])
            elif first_op:
                line = "%s %s %s" % (self.cared_op_layer_names[i - 1], bits[feat_name], bits["image"])
                first_op = False
            elif len(self.net_info[feat_name]["inputs"]) > 0:
                line = " ".join([self.cared_op_layer_names[i - 1], str(bits[feat_name])]
                                + [str(bits[src]) for src in self.net_info[feat_name]["inputs"]])
            else:
                raise NotImplementedError(self.net_info[feat_name])
            lines.append(line)
        if rank == 0:
            with open(table_file, "w") as fh:
                for line in lines:
                    fh.write(line + "\n")
        self.timings = {"pass1_s": t1 - t0, "pass2_s": t2 - t1, "kl_s": t3 - t2, "total_s": time.perf_counter() - t0,
                        "cached_batches": len(cached_ids), "cache_bytes": used, "inplace_consumers": inplace,
                        "fused_bias_absmax_convs": 0 if ctl.fuse_off else sum(1 for m in ctl.fuse_verified if isinstance(m, torch.nn.Conv2d)),
                        "fused_relus": len(ctl.fused_relus),
                        "fused_add_absmax_eltwise": 0 if ctl.fuse_off else sum(1 for m in ctl.fuse_verified if not isinstance(m, torch.nn.Conv2d)),
                        "fused_hist_launches": ctl.hist_fused,
                        "own_conv1x1_launches": ctl.own_conv1x1,
                        "conv_add_chains_proven": len(ctl.defer_ok), "conv_add_launches": ctl.deferred_adds,
                        "conv_add_hist_launches": ctl.deferred_hists,
                        "sums_left_to_pass2_pairs": ctl.pair_sums,
                        "relu_only_chains_proven": len(ctl.relu_only_ok), "launches_without_own_output": ctl.skipped_outputs,
                        "chains_refused_for_keepers": dict(getattr(self, "deferral_refused", None) or {}),
                        "stats_group_bytes": self._stats_limit,
                        "cache_plan": {k: (sorted(v) if isinstance(v, set) else v) for k, v in (plan or {}).items()
                                       if k != "keep"} if plan else None}
        if step_ms:
            self.timings["pass1_step_ms"] = step_ms
        return bits

    def activation_quantize_per_channel(self, images_files, table_file=None):
        """EXTENSION (no reference counterpart): the same two-pass calibration with one histogram row per
        (cared tensor, channel) instead of one per tensor.  Returns {module name: [bit per channel]} and
        writes "<module name> b0 b1 ... b(C-1)" lines to ./workdir/feat_channel.table.  No merge-group
        pooling is applied (the reference defines it for per-tensor scales only); 'image' is included."""
        ChannelCollector = self.channel_collector_cls
        if ChannelCollector is None:
            from common.quantity.channel_collector import ChannelCollector
        rank, _world = _dist_state()
        names = ["image"] + list(self.net_info.keys())
        named_feats, hooks = self.regist_hook_outfeature(self.model)
        ctl = self._hook_ctl
        # the forward's 1x1 convolutions and stem on the own fp32 MFMA kernels (convolution + bias only: the per-channel
        # statistics are taken from the finished tensors)
        patched = self._patch_fused_convs(self.model) if self.device == "gpu" and torch.cuda.is_available() else []
        ctl.own_plain = bool(patched)
        try:
            # One probe forward on a random input of INPUT_SHAPE (what build_net_structure traces with) tells the
            # channel counts and whether later modules overwrite hooked tensors -- on EVERY rank, also one that owns
            # no calibration batch and must still take part in the two all-reduces.  A model in training mode is probed
            # in eval mode (no BatchNorm statistic sees the noise) and then takes one launch per tensor, always correct.
            was_training = self._training_state_modules()
            for m in was_training:
                m.training = False
            try:
                modified = self._probe_forward(ctl.own_plain)
            finally:
                for m in was_training:
                    m.training = True
            collector = ChannelCollector({n: int(named_feats[n].shape[1]) for n in names},
                                         statistic=self.config["SETTINGS"]["STATISTIC"])
            # in-place consumers: one launch per tensor from inside the hooks (the values the reference's hooks would copy)
            self._stats_limit = 0 if (modified or was_training) else _AFTER_FORWARD
            # pass 2 histograms the very tensors pass 1 took the maxima of, for as many batches as fit the allocator's
            # warm pool (whole batches only: _activation_cache_budget; nothing is kept when a later module overwrites
            # hooked tensors in place)
            budget = self._activation_cache_budget() if self._stats_limit else 0
            cached, used = {}, 0
            for _pass in (1, 2):
                fn = collector.refresh_max_val if _pass == 1 else collector.add_to_distributions
                for i, item in self._device_items(images_files):
                    if _pass == 2 and i in cached:
                        fn(cached.pop(i))
                        continue
                    self._forward_with_stats(item, fn, named_feats)
                    if _pass == 1 and budget:
                        need = sum(t.numel() * t.element_size() for t in named_feats.values())
                        if used + need <= budget:
                            cached[i] = dict(named_feats)
                            used += need
                if _dist_on():
                    # pass 1: one MAX all-reduce of fp32[rows]; pass 2: one reduce-scatter -- every rank gets the global histograms of
                    # ITS row block only, sweeps those, and the bits are all-gathered (collector.quantize; _collectives.py)
                    collector.all_reduce_max() if _pass == 1 else collector.reduce_scatter_hist()
                if _pass == 1:
                    collector.intervals()
            bits = collector.quantize()
            self.timings = {"per_channel_cache_bytes": used, "per_channel_kl_s": getattr(collector, "kl_seconds", None)}
        finally:
            ctl.own_plain = False
            for m in patched:
                del m.forward
            for h in hooks:
                h.remove()
            named_feats.clear()
        by_module = OrderedDict()
        by_module["image"] = bits["image"]
        for i, feat_name in enumerate(names[1:]):
            by_module[self.cared_op_layer_names[i]] = bits[feat_name]
        if rank == 0:
            path = table_file or os.path.join(self.config["OUTPUT"]["WORK_DIR"], "feat_channel.table")
            with open(path, "w") as fh:
                for module, b in by_module.items():
                    fh.write(module + " " + " ".join(map(str, b)) + "\n")
        self._channel_collector = collector
        return by_module

    def regist_hook_outfeature(self, model):
        """Forward hooks that expose, after each forward, an ordered dict 'image' + one entry per
        cared node ("<ClassName>_<ordinal>", same keys as net_info) holding the DEVICE tensors.
        Returns (dict, hook handles)."""
        out_feat = OrderedDict()
        handles = []
        cared = set(self.net_info.keys())
        state = {"n": 0}
        total = int(self.layers_num)
        ctl = self._hook_ctl = _HookState()

        def on_forward(module, inputs, output):
            eager = ctl.eager
            if state["n"] == 0:
                out_feat.clear()
                out_feat["image"] = inputs[0].detach()
                if eager is not None:
                    eager.add("image", out_feat["image"])
            state["n"] += 1
            key = "%s_%i" % (type(module).__name__, state["n"])
            if ctl.poison is not None:
                ctl.poison.keys[module] = key
                if ctl.poison.mode == "learn" and torch.is_tensor(output):
                    ctl.poison.outputs[module] = output       # the object the model code holds (out_feat keeps a detach() alias)
            ctl.relu_ready = None        # a ReLU result prepared by the previous module is for the very next forward only
            pending_bias, ctl.fuse_bias = ctl.fuse_bias, None
            fused = pending_bias is not None and self._finish_fused_conv(module, pending_bias, key if key in cared else None, output)
            ctl.last_out = (module, output) if torch.is_tensor(output) else None
            waiting = fused and bool(ctl.deferred) and id(output) in ctl.deferred     # its kernel runs with its Eltwise: noted there
            if key in cared:
                if fused and eager is not None and not ctl.keep_feats:
                    # its statistic is taken and nothing will be cached: do not keep the tensor alive until the end of the
                    # forward (17 GB of references per batch of 256 ResNet-50 images -- in a fresh process that is 30 GB
                    # of allocator pool the forward would not otherwise need, up to a second of hipMalloc)
                    out_feat.pop(key, None)
                    if not waiting:
                        eager.note(key, output)
                else:
                    out_feat[key] = output.detach()
                    if eager is not None:
                        if waiting:
                            pass
                        elif fused:
                            eager.note(key, out_feat[key])
                        else:
                            eager.add(key, out_feat[key])
                if ctl.events is not None:                 # time stamps of one forward, for the cache plan
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    ctl.events.append((state["n"], ev))
            if state["n"] >= total:
                state["n"] = 0
            elif ctl.stop_after is not None and state["n"] >= ctl.stop_after:
                state["n"] = 0
                waiting_convs, ctl.deferred = ctl.deferred, {}
                for d in waiting_convs.values():               # (the forward ends between a convolution and its Eltwise)
                    self._run_deferred(d)
                raise _StopForward()

        for m in model.modules():
            if type(m).__name__ in self._all_op_type:
                handles.append(m.register_forward_hook(on_forward))
        return out_feat, handles

    # ------------------------------------------------------------------------------------------
    # output directories
    # ------------------------------------------------------------------------------------------
    def init_dir(self):
        out = self.config["OUTPUT"]
        for key in ("WORK_DIR", "WEIGHT_DIR", "BIAS_DIR", "FINAL_WEIGHT_DIR", "FINAL_BIAS_DIR"):
            if not os.path.exists(out[key]):
                os.makedirs(out[key], exist_ok=True)

    # ------------------------------------------------------------------------------------------
    # weights
    # ------------------------------------------------------------------------------------------
    def rewrite_weight(self):
        out = self.config["OUTPUT"]
        rank, _ = _dist_state()
        if rank != 0:
            return
        rewriter = BiasReWriter(out["WEIGHT_DIR"], out["BIAS_DIR"], out["FINAL_WEIGHT_DIR"], out["FINAL_BIAS_DIR"],
                                out["WEIGHT_BIT_TABLE"], out["FEAT_BIT_TABLE"],
                                max_shift_limit=self.config["SETTINGS"]["MAX_SHIFT"])
        weight_bits, bias_bits = rewriter.get_weight_info()
        feat_bits, infeat_bits = rewriter.get_feat_info()
        unmatched = set(bias_bits.keys()) ^ set(feat_bits.keys())
        if unmatched:
            print("These layers not include params but we care about their features:", unmatched)
        print("Align bias bit:")
        rewriter.rewrite_bias_table(bias_bits, feat_bits)
        rewriter.rewrite_bias_dir(bias_bits, feat_bits)
        print("Add max shift limitation:")
        need_flag, new_weight = rewriter.max_shift_limit_weight(feat_bits, infeat_bits, weight_bits)
        if need_flag:
            print("rewirte weight!!!!")
            rewriter.rewrite_weight_table(weight_bits, new_weight)
            rewriter.rewrite_weight_dir(weight_bits, new_weight)
        print("Done!")

    def weight_quantize(self):
        """Max-based weight quantisation: bit = 7 - ceil(log2(absmax)), q = clip(rint(w * 2^bit)),
        one JSON file per parameter plus weight.table; then rewrite_weight()."""
        settings = self.config["SETTINGS"]
        out = self.config["OUTPUT"]
        rank, _ = _dist_state()

        names, tensors = [], []
        for name, param in self.model.named_parameters():
            if not name.endswith("weight") and not name.endswith("bias"):
                print("[WARNING]", " not supported param: {}".format(name))
                continue
            owner = self.model
            for part in name.split(".")[:-1]:
                owner = getattr(owner, part)
            t = param.detach()
            if name.endswith("weight") and isinstance(owner, nn.Conv2d):
                if owner.dilation != (1, 1) and not settings["SUPPORT_DILATION"]:
                    t = self.dilation_to_zero_padding(t, owner.dilation)
            names.append(name)
            tensors.append(t.float().contiguous())

        collector = self.collector_cls(names, interval_num=settings["INTERVAL_NUM"], statistic=settings["STATISTIC"],
                                       worker_num=settings["WORKER_NUM"])
        quantizer = self.quantizer_cls(names, worker_num=settings["WORKER_NUM"])
        params = dict(zip(names, tensors))
        collector.refresh_max_val(params)
        max_vals = collector.max_vals
        print("max vals:", max_vals)

        if self._DKL_weight:
            collector.add_to_distributions(params)
            quantizer.quantize(collector.merged_distributions([]), collector.distribution_intervals)
            bits_co = dict(quantizer.bits)
            print("threshold:", quantizer.threshold_value)
        else:
            bits_co = OrderedDict()
            for name in names:
                bits_co[name] = int(8 - 1 - math.ceil(math.log(max_vals[name], 2)))

        table_lines = []
        for name, bit in bits_co.items():
            q = collector.quantize_param(params[name], bit)           # int32 ndarray, clip(rint(w*2^bit))
            table_lines.append(name + " " + str(bit))
            if rank != 0:
                continue
            if name.endswith("weight"):
                dump_int_array(q, os.path.join(out["WEIGHT_DIR"], name + ".json"))
            elif name.endswith("bias"):
                dump_int_array(q, os.path.join(out["BIAS_DIR"], name + ".json"))
            else:
                raise NotImplementedError(name)
        if rank == 0:
            with open(out["WEIGHT_BIT_TABLE"], "w") as fh:
                for line in table_lines:
                    fh.write(line + "\n")
        self.rewrite_weight()

    def dilation_to_zero_padding(self, tensor, dilation):
        """A k x k kernel with dilation 2 as the equivalent dense (2k-1) x (2k-1) kernel."""
        t = torch.as_tensor(tensor)
        assert t.shape[2] == t.shape[3] and tuple(dilation) == (2, 2), "Not support."
        k = t.shape[2]
        dense = torch.zeros(t.shape[0], t.shape[1], 2 * k - 1, 2 * k - 1, dtype=torch.float32, device=t.device)
        dense[..., ::2, ::2] = t
        return dense
