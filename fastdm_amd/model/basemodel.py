"""Checkpoint -> layer loading (mirror of fastdm/model/basemodel.py:5-143).

Same `init_weight(names, dst_module, quant_dtype)` / `weight_loading(model_path | state_dict)`
contract, the same 'every checkpoint tensor must be consumed' assertion, safetensors / .bin /
in-memory state_dict sources.  Tensors go to the GPU first; quantisation runs there.

New here: `weight_loading(..., quantized_cache=path)`.  The reference re-quantises the bf16 checkpoint on
every start-up; with a cache path the first load writes one safetensors file holding every layer exactly as the kernels
consume it (fp8 / int8 weights, scales, int8 column sums, biases, the norm / conv tensors) and later loads read that file
only -- half the bytes of the bf16 checkpoint, no quantisation pass, and the original checkpoint is not needed any more.
"""
import os

import torch

from fastdm_amd.layer.qlinear import QLinear


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
                return self._qc_in.get_tensor(f"t:{name}").to(device=self.device)
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
            dst_module.import_state({k: self._qc_in.get_tensor(f"{key}.{k}") for k in ("w", "s", "c", "b") if f"{key}.{k}" in self._qc_keys})
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
    QC_FORMAT = "fastdm_amd-quantized-v1"

    def weight_loading(self, model_path, data_type=torch.bfloat16, device_type="cuda", quantized_cache=None):
        """basemodel.py:88-143, plus the pre-quantised cache described in the module docstring.  With an existing cache file
        `model_path` may be None."""
        self.origin_tensor_dict = {}
        self.dtype = data_type
        self.device = device_type
        self._qc_in, self._qc_out, self._qc_keys = None, None, frozenset()
        if quantized_cache is not None and os.path.isfile(quantized_cache):
            from safetensors import safe_open
            self._qc_in = safe_open(quantized_cache, framework="pt")
            meta = self._qc_in.metadata() or {}
            if meta.get("format") != self.QC_FORMAT or meta.get("model") != type(self).__name__:
                raise ValueError(f"{quantized_cache}: not a {self.QC_FORMAT} file for {type(self).__name__} ({meta})")
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
                      metadata={"format": self.QC_FORMAT, "model": type(self).__name__})
            os.replace(tmp, quantized_cache)          # readers never see a half-written file
        self._qc_in, self._qc_out, self._qc_keys = None, None, frozenset()
        self._finalize_loading()
        return
