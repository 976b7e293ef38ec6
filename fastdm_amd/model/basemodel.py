"""Checkpoint -> layer loading (mirror of fastdm/model/basemodel.py:5-143).

Same `init_weight(names, dst_module, quant_dtype)` / `weight_loading(model_path | state_dict)`
contract, the same 'every checkpoint tensor must be consumed' assertion, safetensors / .bin /
in-memory state_dict sources.  Tensors go to the GPU first; quantisation runs there.
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
            if name not in self.origin_tensor_dict:
                raise ValueError(f"The {name} is not in origin_tensor_dict!!!")
            t = self.origin_tensor_dict[name].to(device=self.device, dtype=self.dtype).contiguous()
            self.unmatched_tensors.discard(name)
            return t
        if not isinstance(dst_module, QLinear):
            raise ValueError(f"Unsupported dst_module type: {type(dst_module)} weights loading!")
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
        return None

    def _pre_part_loading(self):
        raise NotImplementedError("This method should be implemented in the subclass.")

    def _major_parts_loading(self):
        raise NotImplementedError("This method should be implemented in the subclass.")

    def _post_part_loading(self):
        raise NotImplementedError("This method should be implemented in the subclass.")

    def _finalize_loading(self):
        pass

    def weight_loading(self, model_path, data_type=torch.bfloat16, device_type="cuda"):
        """basemodel.py:88-143."""
        self.origin_tensor_dict = {}
        self.dtype = data_type
        self.device = device_type
        if isinstance(model_path, dict):
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
        self._finalize_loading()
        return
