"""Checkpoint -> layer loading (mirror of fastdm/model/basemodel.py:5-143).

Same `init_weight(names, dst_module, quant_dtype)` / `weight_loading(model_path | state_dict)`
contract, the same 'every checkpoint tensor must be consumed' assertion, safetensors / .bin /
in-memory state_dict sources.  Tensors go to the GPU first; quantisation runs there.

New here: `weight_loading(..., quantized_cache=path)`.  The reference re-quantises the bf16 checkpoint on
every start-up; with a cache path the first load writes one safetensors file holding every layer exactly as the kernels
consume it (fp8 / int8 weights, scales, int8 column sums, biases, the norm / conv tensors) and later loads read that file
only -- half the bytes of the bf16 checkpoint, no quantisation pass, and the original checkpoint is not needed any more.
"""
import hashlib
import os

import torch

from fastdm_amd.layer.qlinear import QLinear


def checkpoint_fingerprint(model_path) -> str:
    """Identity of a checkpoint WITHOUT reading all of its tensor bytes: sorted tensor names, shapes and dtypes, plus
    * files / shard directories: every weight file's path RELATIVE to the checkpoint root and its size, the safetensors header
      (names, shapes, dtypes, data offsets) and 4 KiB samples from the head, middle and tail of EVERY tensor's bytes at its header
      offsets (a partial fine-tune or a merged LoRA keeps layout, header and size but rewrites whole matrices: sampling only the
      ends of the file missed it); .bin files: 64 KiB every 16 MiB.  No mtime and no absolute directory, so a copied
      or re-mounted checkpoint keeps its fingerprint; files are ordered by relative path, so same-named shards in different
      sub-directories (transformer/ and vae/) have a defined order on every filesystem;
    * in-memory state_dicts: the dict's own `fingerprint` attribute when it has one (the lazily generated synthetic
      checkpoints expose their seed that way), otherwise a hash of the first 64 KiB of every tensor.
    Stored in a pre-quantised cache file and compared on every later load that still names a checkpoint, so a cache written
    from other weights (or other arch kwargs of the same class) cannot silently supply them."""
    h = hashlib.sha256()

    def add(*parts):
        for x in parts:
            h.update(str(x).encode())
            h.update(b"\0")

    if isinstance(model_path, dict):
        spec = getattr(model_path, "spec", None)
        if spec is not None:                                  # lazily generated: shapes are known without making tensors
            for k in sorted(spec):
                add(k, tuple(spec[k][0]), spec[k][1])
            add("synthetic", type(model_path).__name__, getattr(model_path, "seed", None))
        else:
            for k in sorted(model_path):
                t = model_path[k]
                add(k, tuple(t.shape), t.dtype)
                flat = t.detach().reshape(-1)[:32768].contiguous().cpu()
                h.update(flat.view(torch.uint8).numpy().tobytes())
        return h.hexdigest()
    import json
    import struct
    model_path = os.path.normpath(model_path)          # 'ckpt/' and 'ckpt' are the same checkpoint
    files = []
    if os.path.isdir(model_path):
        for root, _, names in os.walk(model_path):
            files += [os.path.join(root, n) for n in names if n.endswith((".safetensors", ".bin"))]   # not *.safetensors.index.json
        rel = {f: os.path.relpath(f, model_path).replace(os.sep, "/") for f in files}
    else:
        files = [model_path]
        rel = {model_path: os.path.basename(model_path)}
    SAMPLE = 4096
    for f in sorted(files, key=rel.get):
        size = os.path.getsize(f)
        add(rel[f], size)
        with open(f, "rb") as fh:
            if f.endswith(".safetensors"):
                n = struct.unpack("<Q", fh.read(8))[0]
                header = fh.read(n)
                h.update(hashlib.sha256(header).digest())
                meta = json.loads(header)
                base = 8 + n
                for k in sorted(k for k in meta if k != "__metadata__"):
                    add(k, tuple(meta[k]["shape"]), meta[k]["dtype"])
                    lo, hi = meta[k]["data_offsets"]
                    for off in sorted({lo, max(lo, (lo + hi) // 2 - SAMPLE // 2), max(lo, hi - SAMPLE)}):
                        fh.seek(base + off)
                        h.update(fh.read(min(SAMPLE, hi - off)))
            else:
                for off in range(0, size, 16 << 20):
                    fh.seek(off)
                    h.update(fh.read(65536))
                if size > 65536:
                    fh.seek(size - 65536)
                    h.update(fh.read(65536))
    return h.hexdigest()


class BaseModelCore:
    def __init__(self, type: str = "DiT"):
        self.model_type = type
        self.verbose = False

    def init_weight(self, weights_name, dst_module=None, quant_dtype=None):
        """basemodel.py:9-67."""
        if dst_module is None:  # norm weight
            assert len(weights_name) == 1
            name = weights_name[0]
            if self._qc_in is not None and f"t:{name}" in self._qc_keys:
                self.unmatched_tensors.discard(name)
                t = self._qc_in.get_tensor(f"t:{name}")
                if t.dtype != self.dtype:
                    raise ValueError(f"quantized cache holds {t.dtype} for {name}, the model runs in {self.dtype}: delete the cache")
                return t.to(device=self.device)
            if name not in self.origin_tensor_dict:
                raise ValueError(f"The {name} is not in origin_tensor_dict!!!")
            t = self.origin_tensor_dict[name].to(device=self.device, dtype=self.dtype).contiguous()
            self.unmatched_tensors.discard(name)
            if self._qc_out is not None:
                self._qc_out[f"t:{name}"] = t
            return t
        if not isinstance(dst_module, QLinear):
            raise ValueError(f"Unsupported dst_module type: {type(dst_module)} weights loading!")
        key = "l:" + "+".join(weights_name)
        if self._qc_in is not None and f"{key}.w" in self._qc_keys:
            stored = self._qc_in.get_tensor(f"{key}.w").dtype
            want = quant_dtype if quant_dtype is not None else dst_module.dtype
            if stored != want:
                raise ValueError(f"quantized cache holds {stored} for {key}, the model asks for {want}: delete the cache or change quant_type")
            dst_module.device = self.device
            state = {k: self._qc_in.get_tensor(f"{key}.{k}") for k in ("w", "s", "c", "b") if f"{key}.{k}" in self._qc_keys}
            want_shape = (dst_module.out_features, dst_module.in_features)
            if tuple(state["w"].shape) != want_shape:
                raise ValueError(f"quantized cache holds a {tuple(state['w'].shape)} weight for {key}, the layer is {want_shape} "
                                 "(out_features, in_features): the cache was written for other arch kwargs -- delete it")
            if stored in (torch.float8_e4m3fn, torch.int8) and ("s" not in state or state["s"].numel() != want_shape[0]):
                raise ValueError(f"quantized cache entry {key} lacks its {want_shape[0]} per-channel scales")
            if "b" in state and (state["b"].numel() != want_shape[0] or state["b"].dtype != dst_module.dtype):
                raise ValueError(f"quantized cache entry {key}: bias {tuple(state['b'].shape)} {state['b'].dtype} does not fit the layer")
            dst_module.import_state(state)
            for weight_name in weights_name:
                self.unmatched_tensors.discard(f"{weight_name}.weight")
                self.unmatched_tensors.discard(f"{weight_name}.bias")
            return None
        ws, bs = [], []
        for weight_name in weights_name:
            w_name, b_name = f"{weight_name}.weight", f"{weight_name}.bias"
            if w_name not in self.origin_tensor_dict:
                raise ValueError(f"The {w_name} is not in origin_tensor_dict!!!")
            ws.append(self.origin_tensor_dict[w_name].transpose(0, 1))   # (in_features, out_features) views
            bs.append(self.origin_tensor_dict.get(b_name))
            self.unmatched_tensors.discard(w_name)
            if bs[-1] is not None:
                self.unmatched_tensors.discard(b_name)
        dst_module.device = self.device
        dst_module.weight_loading_and_quant(ws, bs, quant_type=quant_dtype)
        if self._qc_out is not None:
            for k, v in dst_module.export_state().items():
                self._qc_out[f"{key}.{k}"] = v
        return None

    def _pre_part_loading(self):
        raise NotImplementedError("This method should be implemented in the subclass.")

    def _major_parts_loading(self):
        raise NotImplementedError("This method should be implemented in the subclass.")

    def _post_part_loading(self):
        raise NotImplementedError("This method should be implemented in the subclass.")

    def _finalize_loading(self):
        pass

    _qc_in = None
    _qc_out = None
    _qc_keys = frozenset()
    QC_FORMAT = "fastdm_amd-quantized-v3"      # v2: + checkpoint fingerprint and activation dtype in the metadata; v3: the fingerprint samples every tensor

    def weight_loading(self, model_path, data_type=torch.bfloat16, device_type="cuda", quantized_cache=None, verify_against=None):
        """basemodel.py:88-143, plus the pre-quantised cache described in the module docstring.  With an existing cache file
        `model_path` may be None; `verify_against` (the checkpoint the caller would otherwise have loaded) is then only
        fingerprinted and compared with the fingerprint stored in the cache."""
        self.origin_tensor_dict = {}
        self.dtype = data_type
        self.device = device_type
        self._qc_in, self._qc_out, self._qc_keys = None, None, frozenset()
        if quantized_cache is not None and os.path.isfile(quantized_cache):
            from safetensors import safe_open
            self._qc_in = safe_open(quantized_cache, framework="pt")
            meta = self._qc_in.metadata() or {}
            fmt = str(meta.get("format"))
            if fmt != self.QC_FORMAT and fmt.startswith("fastdm_amd-quantized-"):
                raise ValueError(f"{quantized_cache}: stale cache format {fmt} (this build writes {self.QC_FORMAT}): delete the file and "
                                 "re-quantise from the checkpoint")
            if fmt != self.QC_FORMAT or meta.get("model") != type(self).__name__:
                raise ValueError(f"{quantized_cache}: not a {self.QC_FORMAT} file for {type(self).__name__} ({meta})")
            if meta.get("dtype") != str(data_type):
                raise ValueError(f"{quantized_cache} was written for {meta.get('dtype')} activations, the model runs in {data_type}")
            src = verify_against if verify_against is not None else model_path
            if src is not None and (isinstance(src, dict) or os.path.exists(src)):
                fp = checkpoint_fingerprint(src)
                if meta.get("checkpoint") != fp:
                    raise ValueError(f"{quantized_cache} was written from another checkpoint (fingerprint {str(meta.get('checkpoint'))[:12]}..., "
                                     f"this one is {fp[:12]}...): delete the cache file or point quantized_cache elsewhere")
            self._qc_keys = frozenset(self._qc_in.keys())
        elif quantized_cache is not None:
            self._qc_out = {}
        if model_path is None:
            if self._qc_in is None:
                raise ValueError("weight_loading needs a checkpoint (path or state_dict) or an existing quantized_cache file")
        elif isinstance(model_path, dict):
            self.origin_tensor_dict = model_path
        elif os.path.isdir(model_path):
            from safetensors import safe_open
            for root, _, files in os.walk(model_path):
                for name in files:
                    if ".safetensors" in name:
                        with safe_open(os.path.join(root, name), framework="pt") as f:
                            for k in f.keys():
                                self.origin_tensor_dict[k] = f.get_tensor(k)
        elif ".safetensors" in model_path:
            from safetensors import safe_open
            with safe_open(model_path, framework="pt") as f:
                for k in f.keys():
                    self.origin_tensor_dict[k] = f.get_tensor(k)
        else:
            self.origin_tensor_dict = torch.load(model_path, weights_only=True)
        self.unmatched_tensors = set(self.origin_tensor_dict.keys())
        self._pre_part_loading()
        self._major_parts_loading()
        self._post_part_loading()
        if not isinstance(model_path, dict):
            del self.origin_tensor_dict
        assert len(self.unmatched_tensors) == 0, f"unconsumed checkpoint tensors: {sorted(self.unmatched_tensors)[:8]}"
        if self._qc_out is not None:
            from safetensors.torch import save_file
            tmp = f"{quantized_cache}.tmp{os.getpid()}"
            save_file({k: v.detach().contiguous().cpu() for k, v in self._qc_out.items()}, tmp,
                      metadata={"format": self.QC_FORMAT, "model": type(self).__name__, "dtype": str(data_type),
                                "checkpoint": checkpoint_fingerprint(model_path)})
            os.replace(tmp, quantized_cache)          # readers never see a half-written file
        self._qc_in, self._qc_out, self._qc_keys = None, None, frozenset()
        self._finalize_loading()
        return
