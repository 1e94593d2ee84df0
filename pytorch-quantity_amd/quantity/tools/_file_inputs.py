"""The input side of tools.Quantity for file inputs and host tensors (reference quantity/tools/pytorch_quantizer.py:252-284
PRE_PROCESS.IMG 0 / 2 and the loop at :345-426, which feeds ONE file per forward): files grouped into batches, read by a native
reader or a decode pool into a ring of pinned staging buffers, copied to the device on a side stream under the previous batch's
kernels, kept there for pass 2.  Split out of pytorch_quantizer.py in round 4; `Quantity` inherits this mixin (it provides
preprocess(), user_config, device, _calibration_items)."""
import os
import time

import numpy as np
import torch

from common.quantity import _native

__all__ = ["_FileInputs", "_FileGroup"]


class _FileGroup(list):
    """Consecutive calibration files of one rank that go through the model as one batch (Quantity.file_batch)."""


def _dist_on():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


class _FileInputs(object):
    # File inputs (PRE_PROCESS.IMG 0 / 2): the reference feeds ONE file per forward (pytorch_quantizer.py:252-284,
    # 288-296).  Maxima and integer histograms do not depend on how the images are grouped, so `file_batch` consecutive
    # files of a rank go through the model as one batch, decoded by `decode_workers` threads straight into one pinned staging
    # tensor (file_batch = 1: the reference's form).  What the grouping CAN touch is the last bit of some activations: this
    # library's float kernels compute an output element as one fma chain whatever the batch size -- except in the tiles of a
    # launch's partly filled last round, which the tail split (include/fq.h, fq_conv_f32_workspace_bytes) cuts along K, and
    # which tiles those are depends on the launch's tile count.  With _native.conv_tail_split = False (FQ_CONV_TAIL_SPLIT=0)
    # any grouping gives the same histograms bit for bit; with it (default, +3 % images/s) the same feat.table and
    # histograms that differ in a few counts of neighbouring bins (tests/test_gpu_file_inputs.py runs both).
    file_batch = int(os.environ.get("FQ_FILE_BATCH", "64"))
    decode_workers = int(os.environ.get("FQ_DECODE_WORKERS", str(min(16, os.cpu_count() or 8))))
    # pass 2 reads the inputs again: file batches already uploaded in pass 1 stay on the device up to this many bytes
    # (5 120 ResNet images are 3.1 GB) instead of being decoded a second time
    file_keep_bytes = int(float(os.environ.get("FQ_FILE_KEEP_GB", "16")) * (1 << 30))

    def _file_batching(self):
        return (int(self.user_config["PRE_PROCESS"]["IMG"]) in (0, 2) and self.file_batch > 1 and self.device == "gpu"
                and torch.cuda.is_available())

    @staticmethod
    def _npy_header(path):
        """(header bytes up to the data, shape) of an fp32 C-order .npy file, or None."""
        with open(path, "rb") as fh:
            try:
                version = np.lib.format.read_magic(fh)
                shape, fortran, dtype = (np.lib.format.read_array_header_1_0(fh) if version == (1, 0)
                                         else np.lib.format.read_array_header_2_0(fh))
            except ValueError:
                return None
            if fortran or dtype != np.float32:
                return None
            n = fh.tell()
            fh.seek(0)
            return fh.read(n), tuple(shape)

    @staticmethod
    def _read_npy_into(path, dst, header=None):
        """One .npy file (fp32, C order, dst's shape) read straight into `dst` (a numpy view of the pinned batch): no
        intermediate array.  `header`: the header bytes of a file of the same form (Quantity._npy_header) -- files written by
        one np.save loop share them, and comparing bytes is all the parsing the other files need.  False when the file is
        not of that form (the caller falls back to np.load)."""
        if header is None:
            h = _FileInputs._npy_header(path)
            if h is None or h[1] != tuple(dst.shape):
                return False
            header = h[0]
        with open(path, "rb", buffering=0) as fh:
            if fh.read(len(header)) != header:
                return False
            return fh.readinto(memoryview(dst.reshape(-1)).cast("B")) == dst.size * 4

    def _staging(self, shape):
        """A pinned host tensor of `shape` out of a ring of six staging buffers owned by this calibration (allocated on
        first use, as large as the largest batch so far).  Page-locking 150 MB costs 15-40 ms, and torch's caching host
        allocator hands a block back only once the copies out of it are known to be done -- with a batch decoded ahead,
        one being copied and one just consumed it kept allocating new ones.  A slot is reused only after the copy out of it
        has completed (its event, set by _device_items)."""
        ring = self.__dict__.setdefault("_pinned_ring", {"slots": [], "next": 0, "lock": __import__("threading").Lock()})
        with ring["lock"]:
            return self._staging_locked(ring, shape)

    def _staging_locked(self, ring, shape):
        n = 1
        for v in shape:
            n *= int(v)
        if len(ring["slots"]) < 6:
            ring["slots"].append({"buf": torch.empty(n, dtype=torch.float32, pin_memory=True), "event": None})
            slot = ring["slots"][-1]
        else:
            slot = ring["slots"][ring["next"] % 6]
            ring["next"] += 1
            # (three groups are decoded ahead and one is being handed over: at most four slots are out without an event)
            assert slot["event"] is not None or not slot.get("out"), "staging ring too small for the look-ahead"
            if slot["event"] is not None:
                slot["event"].synchronize()
                slot["event"] = None
            if slot["buf"].numel() < n:
                slot["buf"] = torch.empty(n, dtype=torch.float32, pin_memory=True)
        out = slot["buf"][:n].view(shape)
        out._fq_slot = slot
        slot["out"] = True
        return out

    def _preprocess_group(self, group, mode):
        """[G, C, H, W] pinned host tensor of the files in `group` (unreadable image files are dropped, as a crash on one
        would be the reference's only alternative); None when nothing was readable."""
        from concurrent.futures import ThreadPoolExecutor
        if getattr(self, "_decode_pool", None) is None:
            self._decode_pool = ThreadPoolExecutor(max_workers=max(1, self.decode_workers))
        paths = list(group)
        stage = self.__dict__.setdefault("input_wait_s", {})
        t_0 = time.perf_counter()
        if mode == 2:
            h = self._npy_header(paths[0])
            if h is not None:                                   # the whole group through the native reader
                batch = self._staging((len(paths),) + h[1])
                t_1 = time.perf_counter()
                flags = _native.read_npy_batch(paths, h[0], batch, threads=min(4, max(1, self.decode_workers)))
                for j, f in enumerate(flags):
                    if not f:                                   # another header: the general reader, which also checks the shape
                        one = self.preprocess(paths[j])
                        if tuple(one.shape[1:]) != tuple(batch.shape[1:]):
                            raise ValueError("calibration files of one batch have different shapes; set Quantity.file_batch = 1")
                        batch[j].copy_(one[0])
                t_2 = time.perf_counter()
                stage["group_setup"] = stage.get("group_setup", 0.0) + (t_1 - t_0)      # (helper-thread seconds: diagnostics)
                stage["group_read"] = stage.get("group_read", 0.0) + (t_2 - t_1)
                return batch
        first = self.preprocess(paths[0])
        k = 1
        while first is False and k < len(paths):                # (mode 0: skip unreadable files at the head)
            first = self.preprocess(paths[k])
            k += 1
        if first is False or first is None:
            return None
        rest = paths[k:]
        batch = self._staging((1 + len(rest),) + tuple(first.shape[1:]))
        batch[0].copy_(first[0])
        view = batch.numpy()
        t_1 = time.perf_counter()
        header = None
        if mode == 2:
            h = self._npy_header(paths[k - 1])
            header = h[0] if h is not None and h[1] == tuple(view.shape[1:]) else None

        def load(j):
            dst = view[1 + j]
            one = self.preprocess(rest[j])
            if one is False or one is None:
                return False
            if tuple(one.shape[1:]) != tuple(dst.shape):
                return None
            dst[...] = one[0].numpy()
            return True
        # .npy files: one foreign call reads the whole group (fq_read_npy_batch_f32: open / header compare / pread per
        # file on a few host threads, the interpreter lock released throughout).  The same loop in Python measured 17 000-
        # 22 000 files/s alone and 5 500 next to the thread that launches the kernels -- three lock hand-offs per file
        # (scripts/_dbg/file_decode_probe.py); a Python thread pool was slower still.  Files the native reader refuses
        # (another header) and image files (PIL decode + resize: milliseconds each, outside the lock) take load().
        if header is not None and rest:
            flags = _native.read_npy_batch(rest, header, batch[1:], threads=min(4, max(1, self.decode_workers)))
            ok = [True if f else load(j) for j, f in enumerate(flags)]
        elif header is not None:
            ok = []
        else:
            ok = list(self._decode_pool.map(load, range(len(rest))))
        t_2 = time.perf_counter()
        stage["group_setup"] = stage.get("group_setup", 0.0) + (t_1 - t_0)      # (helper-thread seconds: diagnostics)
        stage["group_read"] = stage.get("group_read", 0.0) + (t_2 - t_1)
        if any(r is None for r in ok):
            raise ValueError("calibration files of one batch have different shapes; set Quantity.file_batch = 1")
        keep = [0] + [1 + j for j, r in enumerate(ok) if r]
        if len(keep) == batch.shape[0]:
            return batch
        # unreadable files inside the group: close the gaps IN the staging slot (keep is ascending, so row i <= keep[i] and a
        # front-to-back copy never overwrites a row it still needs) and hand out the head of the same slot.  The view must carry
        # the slot: _device_items records the copy's event on it, and a compacted COPY (what this returned until round 4) left
        # the slot marked "out" with no event -- the ring's seventh group then tripped the look-ahead assertion.
        for i, k_ in enumerate(keep):
            if i != k_:
                batch[i].copy_(batch[k_])
        out = batch[:len(keep)]
        out._fq_slot = batch._fq_slot
        return out

    # Host-resident batches (a DataLoader): the H2D copy of a batch is issued on a side stream from the generator below,
    # i.e. when the loop asks for the NEXT item -- at which point the kernels of the current batch are enqueued but still
    # running -- and the compute stream only waits for the copy's event.  False: plain `.cuda()` on the compute stream.
    prefetch_inputs = True

    def _device_items(self, images_files):
        """(index, network input) for this rank's calibration items; host tensors are copied to the device ahead of the
        compute stream (see prefetch_inputs).  Round 1 measured a helper-THREAD prefetcher with pinned staging buffers as
        slower than a plain `.cuda()` (3 077-3 287 vs 4 232 images/s: the extra host memcpy and GIL traffic outweigh the
        PCIe copy they hide); a side stream needs neither.

        The copy only overlaps the previous batch's kernels for PINNED host tensors (a pageable source makes the copy
        host-synchronous whatever the flag says).  For those nothing on the host waits for the DMA, so the source must
        stay untouched until it is done: the copy's event is waited for before the iterable is asked for its next item --
        a loader that refills one pinned staging buffer per batch would otherwise overwrite a batch still in flight."""
        use_side = (self.prefetch_inputs and self.device == "gpu" and torch.cuda.is_available())
        state = {"in_flight": None}                           # event of a copy whose pinned source is still being read
        items = self._decoded_items(self._calibration_items(images_files))
        kept = getattr(self, "_file_kept", None)
        waits = self.__dict__.setdefault("input_wait_s", {"copy_done": 0.0, "decode": 0.0, "copy_issue": 0.0})

        def fetch():
            """The next readable item with its upload ISSUED: (index, input, event the compute stream must wait for or None);
            None at the end."""
            while True:
                t_a = time.perf_counter()
                if state["in_flight"] is not None:
                    state["in_flight"].synchronize()
                    state["in_flight"] = None
                t_b = time.perf_counter()
                try:
                    i, img, is_group = next(items)
                except StopIteration:
                    return None
                t_c = time.perf_counter()
                waits["copy_done"] += t_b - t_a               # (host seconds this loop spent waiting: diagnostics, Quantity.input_wait_s)
                waits["decode"] += t_c - t_b
                if img is None:                               # a group of unreadable files
                    continue
                done = None
                if use_side and torch.is_tensor(img) and img.device.type != "cuda":
                    if getattr(self, "_copy_stream", None) is None:
                        self._copy_stream = torch.cuda.Stream()
                    with torch.cuda.stream(self._copy_stream):
                        dev = img.cuda(non_blocking=True)
                        done = torch.cuda.Event()
                        done.record(self._copy_stream)
                        slot = getattr(img, "_fq_slot", None)
                        if slot is not None:                  # a staging buffer of this calibration: reused after this event
                            slot["event"] = done
                            slot["out"] = False
                        elif img.is_pinned():
                            state["in_flight"] = done
                    img = dev
                    waits["copy_issue"] += time.perf_counter() - t_c
                    if is_group and kept is not None and i not in kept:
                        nbytes = img.numel() * img.element_size()
                        if self._file_kept_bytes + nbytes <= self.file_keep_bytes:
                            kept[i] = img                     # pass 2 takes the batch from here instead of the files
                            self._file_kept_bytes += nbytes
                return i, img, done

        # One item of look-ahead: the upload of item i + 1 is issued BEFORE the caller enqueues item i's kernels, so it runs
        # beside them whether or not the host is ahead of the device (issued when the caller came back for item i + 1 -- the
        # first form -- it only overlapped as far as the host ran ahead: a 256-image batch is 2.7 ms of PCIe, and that was the
        # file mode's distance to the tensor mode: 20.0 against 17.1 ms per step).  The compute stream waits for an item's own
        # event only, when that item is handed out.
        cur = fetch()
        while cur is not None:
            # (not behind a loader's own pinned buffer: its copy must be DONE before the loader is asked for the next item, and
            #  waiting for that here would stall the host in front of this item's launches)
            nxt = fetch() if (use_side and cur[2] is not None and state["in_flight"] is None) else None
            i, img, done = cur
            if done is not None:
                main = torch.cuda.current_stream()
                main.wait_event(done)
                img.record_stream(main)
            yield i, img
            cur = nxt if nxt is not None else fetch()

    def _decoded_items(self, items):
        """(index, decoded input, is a file group) for every calibration item.  File groups are decoded ONE GROUP AHEAD on
        a helper thread (which fans the files out to the decode pool), so that reading the next batch's files runs beside
        this thread's kernel launches for the current one; a group whose upload was kept in pass 1 is not read again."""
        if not self._file_batching():
            for i, item in items:
                yield i, self.preprocess(item), False
            return
        import collections
        from concurrent.futures import ThreadPoolExecutor
        if getattr(self, "_group_pool", None) is None:
            self._group_pool = ThreadPoolExecutor(max_workers=2)       # two groups in the making: their file reads overlap
        kept = getattr(self, "_file_kept", None) or {}
        pending = collections.deque()
        it = iter(items)

        def refill():
            for i, item in it:
                pending.append((i, None if i in kept else self._group_pool.submit(self.preprocess, item)))
                return
        refill()
        refill()
        refill()
        while pending:
            i, fut = pending.popleft()
            refill()
            yield i, (kept[i] if fut is None else fut.result()), True
