"""Inference-side counterpart of the reference's dataset plumbing:

  * `SPLITS` / `load_pairs_json`   ~ data/datasets/builtin.py:15-51, data/datasets/mp3d.py:18-45 (the json's "data" list)
  * `PairMapper`                   ~ data/planercnn_transforms.py:205-398 restricted to what inference reads: the two images of
    a pair as float32 [3,H,W] tensors in cfg.INPUT.FORMAT channel order with values 0..255, file names / ids, `rel_pose`.
    Matterport3D images are used as stored (:210-227); ScanNet images are resized to 640 x 480 (:312-314, cv2.resize =
    8-bit INTER_LINEAR) - here on the GPU by `nopesac_resize_bilinear_u8` (csrc/resize.hip).
Decoding: baseline JPEG files (the ScanNet colour frames) are decoded ON THE GPU when one is present (nopesac_amd/jpeg.py,
csrc/jpeg.hip - bit-exact with the PIL / libjpeg-turbo decode detectron2's `utils.read_image` performs); PNG files (Matterport3D) and
JPEG variants outside the decoder's subset (progressive, CMYK, ...) are decoded by PIL on the host, as the reference decodes every
file.  Ground-truth masks / depth / k-means pose classes are not read: nothing on the inference path consumes them.
"""
from __future__ import annotations

import json
import os
from typing import List

import numpy as np
import torch

SPLITS = {   # builtin.py:15-21: name -> (root folder under ./datasets, annotation json)
    "mp3d_val": ("mp3d_dataset", "mp3d_planercnn_json/cached_set_val.json"),
    "mp3d_test": ("mp3d_dataset", "mp3d_planercnn_json/cached_set_test.json"),
    "mp3d_train": ("mp3d_dataset", "mp3d_planercnn_json/cached_set_train.json"),
    "scannet_train": ("scannet_dataset", "scannet_json/cached_set_trainV2.json"),
    "scannet_test": ("scannet_dataset", "scannet_json/cached_set_testV2.json"),
}
MP3D_ORIGINAL_ROOT = "/Pool1/users/jinlinyi/dataset/mp3d_rpnet_v4_sep20/"   # prefix stored in the json (planercnn_transforms.py:213-214)


def dataset_json(name: str, datasets_dir: str = "./datasets") -> str:
    if name not in SPLITS:
        raise KeyError(f"unknown dataset {name!r}; known: {sorted(SPLITS)}")
    root, rel = SPLITS[name]
    return os.path.join(datasets_dir, root, rel)


def load_pairs_json(json_file: str) -> List[dict]:
    """The reference's `load_mp3d_json`: the json's "data" list (one dict per image pair, no pixels)."""
    with open(json_file, "r") as f:
        summary = json.load(f)
    if "data" not in summary:
        raise ValueError(f"{json_file}: not a NopeSAC pair file (no 'data' list)")
    return summary["data"]


def read_png_native(blob: bytes, fmt: str = "BGR"):
    """A PNG file's pixels as uint8 [H,W,3] through the library's host decoder (csrc/png_host.hip: zlib inflate + row filters + the
    mode conversion of PIL's convert("RGB"); called with the interpreter lock RELEASED - PIL holds it while it decodes a PNG, which capped
    the reader threads at ~90 images/s whatever their number).  None when the file is a variant the decoder leaves to PIL (16-bit,
    sub-byte, interlaced), is corrupt, or the library is not built."""
    import ctypes
    try:
        from . import _lib
        L = _lib.load()
    except (RuntimeError, OSError, AttributeError):
        return None
    h, w, ch, ok = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    if L.nopesac_png_info_host(blob, len(blob), ctypes.byref(h), ctypes.byref(w), ctypes.byref(ch), ctypes.byref(ok)) != 0 or not ok.value:
        return None
    out = np.empty((h.value, w.value, 3), np.uint8)
    rc = L.nopesac_png_decode_host(blob, len(blob), out.ctypes.data, out.size, 1 if fmt == "BGR" else 0)
    return out if rc == 0 else None


def read_png_files(paths: List[str], fmt: str = "BGR", height: int = 480, width: int = 640, chw: bool = True, threads: int = 4, out: torch.Tensor = None):
    """A batch of PNG files decoded by ONE library call on `threads` threads of its own (csrc/png_host.hip
    nopesac_png_decode_files_host: no per-image interpreter work - with one ctypes call per image the open / read / allocate / transpose
    steps around it, interpreter lock held, capped 32 reader threads at 6.1 k images/s on cores that inflate 11 k).  Returns (uint8 tensor
    [n, 3, H, W] (chw) or [n, H, W, 3], status list): status[i] != 0 -> image i was NOT written (another geometry, a variant the decoder
    leaves to PIL, unreadable) and the caller reads that file itself.  `out`: a (pinned) uint8 buffer of that shape to decode into."""
    import ctypes
    from . import _lib
    L = _lib.load()
    n = len(paths)
    shape = (n, 3, height, width) if chw else (n, height, width, 3)
    if out is None:
        out = torch.empty(shape, dtype=torch.uint8)
    if tuple(out.shape) != shape or out.dtype != torch.uint8 or not out.is_contiguous() or out.device.type != "cpu":
        raise ValueError("read_png_files: out must be a contiguous host uint8 tensor of shape %r" % (shape,))
    if n == 0:
        return out, []
    status = (ctypes.c_int * max(1, n))()
    arr = (ctypes.c_char_p * max(1, n))(*[os.fsencode(p) for p in paths])
    rc = L.nopesac_png_decode_files_host(arr, n, out.data_ptr(), 3 * height * width, height, width, (1 if fmt == "BGR" else 0) | (2 if chw else 0),
                                         max(1, int(threads)), status)
    if rc < 0:
        raise RuntimeError("nopesac_png_decode_files_host: bad arguments")
    return out, list(status[:n])


def read_image(path: str, fmt: str = "BGR") -> np.ndarray:
    """detectron2 `utils.read_image`: uint8 [H,W,3] in `fmt` order (EXIF orientation ignored like d2 v0.4).  PNG files (the mp3d
    split) go through `read_png_native` when it takes them (bit-identical pixels: tests/test_host_cpu.py); everything else through PIL."""
    if fmt not in ("BGR", "RGB"):
        raise ValueError(f"unsupported INPUT.FORMAT {fmt!r}")
    if path.lower().endswith(".png") and os.environ.get("NOPESAC_PNG_NATIVE", "1") != "0":
        with open(path, "rb") as f:
            blob = f.read()
        arr = read_png_native(blob, fmt)
        if arr is not None:
            return arr
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"))
    if fmt == "BGR":
        arr = arr[:, :, ::-1]
    elif fmt != "RGB":
        raise ValueError(f"unsupported INPUT.FORMAT {fmt!r}")
    return np.ascontiguousarray(arr)


def _png_batch_available() -> bool:
    try:
        from . import _lib
        return hasattr(_lib.load(), "nopesac_png_decode_files_host")
    except (RuntimeError, OSError, AttributeError):
        return False


def is_jpeg_path(path: str) -> bool:
    return path.lower().endswith((".jpg", ".jpeg"))


class PairMapper:
    """dataset dict -> model input dict (the `image` tensors stay on the host unless `device` is given, exactly like the
    reference mapper's output; ScanNet images pass through the GPU resize kernel)."""

    def __init__(self, cfg, dataset_name: str = "", device=None, uint8: bool = False, gpu_jpeg: bool = None):
        """uint8: keep the decoded 8-bit samples (uint8 CHW tensors; the model widens them on the device - bit-identical results,
        a quarter of the host-to-device bytes) instead of the reference mapper's float32 tensors.
        gpu_jpeg: decode baseline JPEG files on the GPU (default: whenever a GPU is present); files the decoder does not support go
        through PIL like every file does in the reference (counted in `self.host_decoded`)."""
        self.uint8 = uint8
        self.gpu_jpeg = gpu_jpeg
        self.host_decoded = 0
        self.img_format = cfg.INPUT.FORMAT
        self.root_dir = cfg.DATASETS.ROOT_DIR
        name = dataset_name or (cfg.DATASETS.TEST[0] if len(cfg.DATASETS.TEST) else "")
        self.scannet = "scannet" in name
        self.device = device
        # decoder threads start with current device 0: remember the device of the thread that builds the mapper (the rank's)
        self._resize_device = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                                                 if torch.cuda.is_available() else None)

    def _use_gpu_jpeg(self) -> bool:
        if self.gpu_jpeg is None:                        # NOPESAC_GPU_JPEG=0: decode every file with PIL on the host (A/B runs)
            self.gpu_jpeg = self._resize_device is not None and os.environ.get("NOPESAC_GPU_JPEG", "1") != "0"
        return bool(self.gpu_jpeg) and self._resize_device is not None

    def _finish_device_image(self, img_t: torch.Tensor, keep_device: bool = False) -> torch.Tensor:
        """uint8 [H,W,3] on the GPU (cfg.INPUT.FORMAT order) -> what _image returns (resized for ScanNet, CHW, uint8 / float32, on
        self.device or the host)"""
        if self.scannet and tuple(img_t.shape[:2]) != (480, 640):
            from . import ops
            img_t = ops.resize_bilinear_u8(img_t.contiguous(), 480, 640)
        t = img_t.permute(2, 0, 1).contiguous() if self.uint8 else img_t.permute(2, 0, 1).float()
        return t if (self.device is not None or keep_device) else t.cpu()

    def _finish_device_batch(self, imgs: List[torch.Tensor], keep_device: bool) -> List[torch.Tensor]:
        """_finish_device_image for the images of one decode_batch call.  When they share one size (a ScanNet batch: 968 x 1296) the
        resize + CHW transpose of the WHOLE batch is one launch (per image: resize, permute, contiguous = three; 192 launches per batch of 32
        pairs on the thread that also launches the model); the results are views of one [n,3,480,640] tensor.  keep_device: GPU-decoded
        images stay in HBM although the mapper was built without a device (LazyPairs' GPU path: the model reads them in place - the host
        round trip `.cpu()` + the model's own H2D copy was 17 of the 22 ms a batch took to map)."""
        from . import ops
        batch = None
        if len(imgs) > 1 and len({tuple(t.shape) for t in imgs}) == 1:
            if self.scannet and tuple(imgs[0].shape[:2]) != (480, 640):
                batch = ops.resize_bilinear_u8_batch(imgs, 480, 640, chw=True)
        if batch is None:
            return [self._finish_device_image(t, keep_device) for t in imgs]
        if not self.uint8:
            batch = batch.float()
        if self.device is None and not keep_device:
            batch = batch.cpu()
        return [batch[i] for i in range(len(imgs))]

    def decode_files(self, paths: List[str], blobs: List[bytes] = None, infos: list = None, sync: bool = True, keep_device: bool = False,
                     host=None) -> List[torch.Tensor]:
        """The images of `paths` as _image() returns them, the JPEG files among them decoded in ONE launch chain on the GPU (all
        restart intervals / images of the batch in flight together).  blobs / infos: file contents and jpeg.parse results when the
        caller (LazyPairs' reader threads) has them already.  sync (default): device tensors are COMPLETE when this returns - the
        decode ran on the calling thread's current stream and whoever consumes the images may use any other stream; sync = False only
        for a caller that records its own event behind the decode (LazyPairs._iter_batches_gpu)."""
        from . import jpeg
        dev = self._resize_device
        if host is not None and host.n == len(paths) and self._use_gpu_jpeg():       # the whole batch prepared already (reader thread, natively)
            with torch.cuda.device(dev):
                dec = jpeg.decode_batch(None, dev, bgr=(self.img_format == "BGR"), host=host)
                out = self._finish_device_batch(dec, keep_device)
                if sync and (self.device is not None or keep_device):
                    torch.cuda.current_stream(dev).synchronize()
            return out
        blobs = list(blobs) if blobs is not None else [None] * len(paths)
        infos = list(infos) if infos is not None else [None] * len(paths)
        gpu_idx = []
        for i, p in enumerate(paths):
            if not (self._use_gpu_jpeg() and is_jpeg_path(p)):
                continue
            if blobs[i] is None:
                with open(p, "rb") as f:
                    blobs[i] = f.read()
            if infos[i] is None:
                try:
                    infos[i] = jpeg.parse(blobs[i])
                except jpeg.JpegUnsupported:
                    infos[i] = False
            if infos[i]:
                gpu_idx.append(i)
        out = [None] * len(paths)
        if gpu_idx:
            with torch.cuda.device(dev):
                dec = jpeg.decode_batch([blobs[i] for i in gpu_idx], dev, bgr=(self.img_format == "BGR"), infos=[infos[i] for i in gpu_idx])
                for i, t in zip(gpu_idx, self._finish_device_batch(dec, keep_device)):
                    out[i] = t
                if sync and (self.device is not None or keep_device):          # (host hand-over: .cpu() above has synchronised already)
                    torch.cuda.current_stream(dev).synchronize()
        for i, p in enumerate(paths):
            if out[i] is None:
                if is_jpeg_path(p) and self._use_gpu_jpeg():
                    self.host_decoded += 1
                out[i] = self._image_host(p)
        return out

    def _image(self, path: str) -> torch.Tensor:
        if self._use_gpu_jpeg() and is_jpeg_path(path):
            return self.decode_files([path])[0]
        return self._image_host(path)

    def _image_host(self, path: str) -> torch.Tensor:
        img = read_image(path, self.img_format)
        if self.scannet and img.shape[:2] != (480, 640):
            from . import ops
            dev = self._resize_device
            with torch.cuda.device(dev):
                img_t = ops.resize_bilinear_u8(torch.from_numpy(img.copy()).to(dev), 480, 640)
            t = img_t.permute(2, 0, 1).contiguous() if self.uint8 else img_t.permute(2, 0, 1).float()
            return t if self.device is not None else t.cpu()
        t = torch.as_tensor(np.ascontiguousarray(img.transpose(2, 0, 1)) if self.uint8 else img.transpose(2, 0, 1).astype("float32"))
        return t.to(self.device) if self.device is not None else t

    def __call__(self, dataset_dict: dict, images: List[torch.Tensor] = None) -> dict:
        """images: the two decoded images when the caller decoded a whole batch at once (map_batch)"""
        # (the reference mapper deep-copies the dict because it edits the annotations; here only these four keys of the two view dicts are
        #  set, so the view dicts are copied and everything below them - the annotation lists of a pair are kilobytes - is shared: a deep
        #  copy per pair is interpreter time the loader threads take from the thread that launches the model)
        d = dict(dataset_dict)
        for v in "01":
            d[v] = dict(dataset_dict[v])
        for vi, v in enumerate("01"):
            if not self.scannet and self.root_dir:
                d[v]["file_name"] = d[v]["file_name"].replace(MP3D_ORIGINAL_ROOT, self.root_dir)
            d[v]["image"] = images[vi] if images is not None else self._image(d[v]["file_name"])
            h, w = d[v]["image"].shape[-2:]
            if "height" in d[v] and (d[v]["height"], d[v]["width"]) != (h, w) and not self.scannet:
                raise ValueError(f"{d[v]['file_name']}: image is {h}x{w}, annotation says {d[v]['height']}x{d[v]['width']}")
            d[v]["height"], d[v]["width"] = h, w
        return d


    def file_names(self, dataset_dict: dict) -> List[str]:
        names = [dataset_dict[v]["file_name"] for v in "01"]
        if not self.scannet and self.root_dir:
            names = [n.replace(MP3D_ORIGINAL_ROOT, self.root_dir) for n in names]
        return names

    def batch_paths(self, entries: List[dict]) -> List[str]:
        """The 2 * len(entries) files of a batch in the MODEL's order - every pair's view 0, then every pair's view 1: images decoded
        into one buffer in this order reach the model's input tensor in one copy (PlaneTR_NopeSAC._as_one_batch)."""
        names = [self.file_names(e) for e in entries]
        return [n[0] for n in names] + [n[1] for n in names]

    def map_batch(self, entries: List[dict], blobs: List[bytes] = None, infos: list = None, sync: bool = True, keep_device: bool = False,
                  host=None) -> List[dict]:
        """The mapped dicts of a batch of pairs with all 2 * len(entries) images decoded together (decode_files; blobs / infos in
        batch_paths order)."""
        B = len(entries)
        imgs = self.decode_files(self.batch_paths(entries), blobs, infos, sync=sync, keep_device=keep_device, host=host)
        return [self(e, images=[imgs[i], imgs[B + i]]) for i, e in enumerate(entries)]


class LazyPairs:
    """The pairs of a dataset split, decoded and mapped ON DEMAND by a small pool of threads that runs ahead of the consumer (the
    reference: a torch DataLoader with DATALOADER.NUM_WORKERS worker processes, batch size 1).  Materialising a split up front is
    7.4 MB of float32 pixels per pair (1.8 MB as uint8) and one PIL decode after the other; here at most `prefetch` mapped pairs exist
    at a time.  Sequence protocol: len(), integer index (mapped dict), slice (another LazyPairs over the same json entries: rank
    shards), iteration and iter_batches() in order."""

    def __init__(self, entries: List[dict], mapper: "PairMapper", workers: int = 4, prefetch: int = 64):
        import threading
        self.entries, self.mapper = entries, mapper
        self.workers, self.prefetch = max(1, int(workers)), max(1, int(prefetch))
        self._decode_lock = threading.Lock()

    def __len__(self) -> int:
        return len(self.entries)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return LazyPairs(self.entries[idx], self.mapper, self.workers, self.prefetch)
        return self.mapper(self.entries[idx])

    def __iter__(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        if not self.entries:
            return
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            pending, nxt = deque(), 0
            while nxt < len(self.entries) and len(pending) < self.prefetch:
                pending.append(pool.submit(self.mapper, self.entries[nxt]))
                nxt += 1
            while pending:
                item = pending.popleft().result()
                if nxt < len(self.entries):
                    pending.append(pool.submit(self.mapper, self.entries[nxt]))
                    nxt += 1
                yield item

    def _read_batch(self, entries: List[dict]):
        """reader thread: the host side of a batch's JPEG files.  Fast path: ONE library call (jpeg.prepare_files, csrc/jpeg_host.hip: read,
        marker walk, stuffing removal, tables, launch arrays on a few threads of its own, no interpreter).  If it reports a file it does not
        take: file contents + jpeg.parse per file in Python, which sorts out what PIL must decode."""
        from . import jpeg
        paths = self.mapper.batch_paths(entries)
        if paths and all(is_jpeg_path(p) for p in paths) and os.environ.get("NOPESAC_JPEG_NATIVE_PREPARE", "1") != "0":
            host, status = jpeg.prepare_files(paths, threads=max(1, self.workers // 4))
            if host is not None:
                return entries, None, None, host
        blobs, infos = [None] * len(paths), [None] * len(paths)
        for i, p in enumerate(paths):
            if is_jpeg_path(p):
                with open(p, "rb") as f:
                    blobs[i] = f.read()
                try:
                    infos[i] = jpeg.parse(blobs[i])
                except jpeg.JpegUnsupported:
                    infos[i] = False
        host = jpeg.prepare_batch(infos) if (infos and all(infos)) else None
        return entries, blobs, infos, host

    def _iter_batches_gpu(self, pairs_per_batch: int, ahead: int = 0):
        """JPEG splits with a GPU: reader threads load and parse the files of whole batches ahead of the consumer; this thread
        launches each batch's decode (one chain for its 2 * pairs_per_batch images) on one of `ahead` decode streams, up to `ahead`
        batches (default 6, NOPESAC_JPEG_AHEAD: under the model's four batches in flight a decode chain waits its turn on the GPU for
        tens of milliseconds - measured end to end on 968 x 1296 frames: 2060 / 2430 / 2170 pairs/s with 3 / 6 / 10 ahead)
        before the consumer needs them - the serial Huffman chains of different batches run next to each other and next to
        the model; the consumer's stream waits for the batch's event."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        dev = self.mapper._resize_device
        ahead = ahead or max(1, int(os.environ.get("NOPESAC_JPEG_AHEAD", "6")))
        chunks = [self.entries[i:i + pairs_per_batch] for i in range(0, len(self.entries), pairs_per_batch)]
        with torch.cuda.device(dev):
            streams = [torch.cuda.Stream(device=dev) for _ in range(ahead)]
        # A decode THREAD launches the chains (a dozen torch calls, seven host-to-device copies and ten kernel launches per batch: ~3 ms of
        # wall time that the consumer's thread - which launches the model and packages its results - does not have); the consumer only
        # waits for a finished batch.  NOPESAC_JPEG_DECODE_THREAD=0: launch from the consumer's thread (the round-4 form).
        import queue
        import threading
        done_q: "queue.Queue" = queue.Queue(maxsize=ahead)
        stop = threading.Event()
        depth = max(ahead + 1, self.prefetch // max(1, pairs_per_batch))

        def launch(k, res):
            entries, blobs, infos, host = res
            st = streams[k % ahead]
            with torch.cuda.device(dev), torch.cuda.stream(st):
                items = self.mapper.map_batch(entries, blobs, infos, sync=False, keep_device=True, host=host)
                ev = torch.cuda.Event()
                ev.record()
            return items, ev

        def put(x):                                             # (gives up when the consumer has gone away)
            while not stop.is_set():
                try:
                    done_q.put(x, timeout=0.1)
                    return
                except queue.Full:
                    pass

        def decoder(pool):
            try:
                reading, nxt, k = deque(), 0, 0
                while (nxt < len(chunks) or reading) and not stop.is_set():
                    while nxt < len(chunks) and len(reading) < depth:
                        reading.append(pool.submit(self._read_batch, chunks[nxt]))
                        nxt += 1
                    put(launch(k, reading.popleft().result()))
                    k += 1
                put(None)
            except BaseException as e:                          # surfaces in the consumer
                put(e)

        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            if os.environ.get("NOPESAC_JPEG_DECODE_THREAD", "1") == "0":
                reading, decoding, nxt, k = deque(), deque(), 0, 0
                while nxt < len(chunks) or reading or decoding:
                    while nxt < len(chunks) and len(reading) + len(decoding) < depth:
                        reading.append(pool.submit(self._read_batch, chunks[nxt]))
                        nxt += 1
                    while reading and len(decoding) < ahead:
                        decoding.append(launch(k, reading.popleft().result()))
                        k += 1
                    items, ev = decoding.popleft()
                    ev.synchronize()
                    yield items
                return
            th = threading.Thread(target=decoder, args=(pool,), name="nopesac-jpeg-decode", daemon=True)
            th.start()
            try:
                while True:
                    item = done_q.get()
                    if item is None:
                        break
                    if isinstance(item, BaseException):
                        raise item
                    items, ev = item
                    # The consumer picks its own stream AFTER next() returns (run.inference_on_dataset rotates side streams), so a
                    # wait_event on this thread's current stream would order nothing: the batch is handed out COMPLETE.  The decode was
                    # launched up to `ahead` batches ago - this wait is normally over already.  (Memory: the images were allocated on the
                    # decode stream; PlaneTR_NopeSAC._copy_images records the consumer's stream on every device image it reads.)
                    ev.synchronize()
                    yield items
            finally:
                stop.set()
                th.join()

    def _png_batch(self, entries: List[dict]):
        """batch thread: every image of the batch through ONE native call (read_png_files, `workers` threads), files it does not take
        through read_image; the mapped dicts hold views of the batch tensor (uint8 CHW) or of its float32 copy."""
        m = self.mapper
        paths = m.batch_paths(entries)                            # the model's batch order: its H2D copy of the batch is then ONE
        B = len(entries)                                          # transfer of this buffer
        H, W = int(entries[0]["0"].get("height", 480)), int(entries[0]["0"].get("width", 640))          # (files of another size: status -5)
        # The batch buffer comes from torch's PINNED host allocator when there is a GPU: it caches freed blocks, so after the first few
        # batches a buffer is a recycled one - already faulted in (decoding into fresh pageable memory is bound by its page faults:
        # measured 2.2 k images/s on 32 threads against 9.9 k into a buffer that has been touched before), the H2D copy of its images is a
        # DMA, and the allocator itself keeps a block from being reused while such a copy is in flight.
        pin = torch.cuda.is_available()
        buf = torch.empty((len(paths), 3, H, W), dtype=torch.uint8, pin_memory=pin)
        with self._decode_lock:                                   # one decode at a time, on all of this loader's threads; the mapping of
            buf, status = read_png_files(paths, m.img_format, H, W, chw=True, threads=self.workers, out=buf)      # the batch before overlaps it
        imgs = [None] * len(paths)
        for i, st in enumerate(status):
            if st != 0:                                           # another size / colour depth / not a PNG after all: the general reader
                imgs[i] = m._image_host(paths[i])
        batch = buf if m.uint8 else torch.empty(buf.shape, dtype=torch.float32, pin_memory=pin).copy_(buf)
        if m.device is not None:
            batch = batch.to(m.device, non_blocking=False)
        for i in range(len(paths)):
            if imgs[i] is None:
                imgs[i] = batch[i]
        return [m(e, images=[imgs[i], imgs[B + i]]) for i, e in enumerate(entries)]

    def _iter_batches_png(self, pairs_per_batch: int, ahead: int = 2):
        """PNG splits (mp3d): whole batches decoded `ahead` of the consumer, each by one native call that runs on `workers` threads
        without the interpreter (two batch threads: one batch's mapping overlaps the next one's decode)."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        chunks = [self.entries[i:i + pairs_per_batch] for i in range(0, len(self.entries), pairs_per_batch)]
        with ThreadPoolExecutor(max_workers=2) as pool:
            pending, nxt = deque(), 0
            while nxt < len(chunks) or pending:
                while nxt < len(chunks) and len(pending) < ahead + 1:
                    pending.append(pool.submit(self._png_batch, chunks[nxt]))
                    nxt += 1
                yield pending.popleft().result()

    def iter_batches(self, pairs_per_batch: int):
        if self.mapper._use_gpu_jpeg() and self.entries and is_jpeg_path(self.mapper.file_names(self.entries[0])[0]):
            yield from self._iter_batches_gpu(pairs_per_batch)
            return
        if (self.entries and not self.mapper.scannet and self.mapper.file_names(self.entries[0])[0].lower().endswith(".png")
                and os.environ.get("NOPESAC_PNG_NATIVE", "1") != "0" and _png_batch_available()):
            yield from self._iter_batches_png(pairs_per_batch)
            return
        batch = []
        for item in self:
            batch.append(item)
            if len(batch) == pairs_per_batch:
                yield batch
                batch = []
        if batch:
            yield batch


def build_inference_pairs(cfg, dataset_name: str, datasets_dir: str = "./datasets", limit: int = 0, device=None, uint8: bool = False,
                          lazy: bool = False, prefetch: int = 64):
    """The split as a list of mapped input dicts - or, `lazy`, as a LazyPairs that decodes ahead of the consumer with
    cfg.DATALOADER.NUM_WORKERS threads."""
    pairs = load_pairs_json(dataset_json(dataset_name, datasets_dir))
    if limit:
        pairs = pairs[:limit]
    mapper = PairMapper(cfg, dataset_name, device, uint8=uint8)
    if lazy:
        return LazyPairs(pairs, mapper, workers=int(getattr(cfg.DATALOADER, "NUM_WORKERS", 4)) or 1, prefetch=prefetch)
    return [mapper(p) for p in pairs]
