"""Rebuild a calibrated model as its integer-simulation (ReconModel) or fake-quant (ReconTest)
counterpart by swapping Conv2d / Linear / Eltwise modules for the quantize-op modules.

Drop-in for reference quantity/tools/reconstruction.py (Reconstruction :93, load_configs :101-105,
get_quantity_information :107-172, ReconModel :175-241, ReconTest :243-324, merge_bn :326-332):
same class and method names, same per-layer info dict, same torch.save of the whole model (so the
pickled class paths common.quantity.new_quantity_op.* stay loadable).  The dead
Run_model_quantizer helper of the reference (:18-89, imports modules that do not exist) is not
reproduced.
"""
from collections import OrderedDict

import torch
import yaml

from common.quantity import BitReader, Eltwise, Concat, Identity  # noqa: F401
from common.quantity import NewConv2d, NewAdd, NewLinear, TestConv, TestLinear
from common.quantity import merge_bn

__all__ = ["Reconstruction"]


def _swap(root, dotted_name, replacement):
    parent = root
    parts = dotted_name.split(".")
    if not parts or not parts[-1]:
        raise ValueError("the layer name is wrong")
    for p in parts[:-1]:
        parent = getattr(parent, p)
    parent.add_module(parts[-1], replacement)


class Reconstruction(object):

    def __init__(self, model):
        super(Reconstruction, self).__init__()
        self.model = model
        self.load_configs()

    def load_configs(self):
        with open("../tools/configs.yml") as fh:
            self.config = yaml.safe_load(fh)

    def merge_bn(self):
        return merge_bn(self.model)

    def get_quantity_information(self):
        """{layer name: {weight_bit, bias_bit, output_bit, input_bit, layer, layer_type}} from
        weight.table and feat.table.  bias_bit is the layer's OUTPUT bit (the aligned bias), the
        input bit is the first input's bit; parameter-free cared layers (image, Eltwise, Concat)
        get weight_bit = bias_bit = None."""
        cared = self.config["SETTINGS"]["CARE_OP_TYPE"]
        reader = BitReader(feat_table=self.config["OUTPUT"]["FEAT_BIT_TABLE"],
                           weight_table=self.config["OUTPUT"]["WEIGHT_BIT_TABLE"])
        weight_bits, bias_bits = reader.get_weight_info()
        feat_bits, infeat_bits = reader.get_feat_info()
        info = OrderedDict()
        for name, wbit in weight_bits.items():
            assert name in feat_bits, "{} not in {}".format(name, feat_bits)
            assert name in infeat_bits, "{} not in {}".format(name, infeat_bits)
            out_bit = feat_bits[name]
            in_bit = int(infeat_bits[name][0])
            print("name: {} weight:{} bias:{} in:{} out:{}".format(name, wbit, bias_bits[name], in_bit, out_bit))
            if name not in info:
                info[name] = {"weight_bit": wbit, "bias_bit": out_bit, "output_bit": out_bit, "input_bit": in_bit}
        for name, bit in feat_bits.items():
            if name in info:
                continue
            info[name] = {"weight_bit": None, "bias_bit": None, "output_bit": bit,
                          "input_bit": None if name == "image" else int(infeat_bits[name][0])}
        for name, module in self.model.named_modules():
            kind = type(module).__name__
            if name in info and kind in cared:
                info[name]["layer"] = module
                info[name]["layer_type"] = kind
        return info

    def _rebuild(self, all_quantize_infor, new_model_path, make_conv, make_linear, label):
        for name, module in list(self.model.named_modules()):
            kind = type(module).__name__
            if kind not in ("Conv2d", "Linear", "Eltwise"):
                continue
            assert all_quantize_infor[name]["layer_type"] == kind, "layer type wrong"
            if kind == "Conv2d":
                new = make_conv(name, module, all_quantize_infor[name])
            elif kind == "Linear":
                new = make_linear(name, module, all_quantize_infor[name])
            else:
                new = NewAdd()
            _swap(self.model, name, new)
            print("The layer change: {} ==>{} ".format(name, type(new).__name__))
        print("Model reconstruction successfully !")
        torch.save(self.model, new_model_path)
        return self.model

    def ReconModel(self, all_quantize_infor, new_model_path):
        """Quantity -> integer conv/linear -> RightShift -> BiasAdd -> Sp -> DeQuantity per layer."""
        return self._rebuild(all_quantize_infor, new_model_path,
                             lambda n, m, q: NewConv2d(m, q), lambda n, m, q: NewLinear(m, q), "ReconModel")

    def make_resident(self, example_input, verify=True):
        """Extension: keep the integer-simulation model's activations as integers between its layers
        (common.quantity.resident.enable on self.model, which must already be the ReconModel on the GPU).
        Same outputs bit for bit; returns the plan summary."""
        from common.quantity import resident
        return resident.enable(self.model, example_input, verify=verify)

    def ReconTest(self, all_quantize_infor, new_model_path):
        """Fake-quant evaluation model: w, b fake-quantised once; every output fake-quantised."""
        return self._rebuild(all_quantize_infor, new_model_path,
                             lambda n, m, q: TestConv(n, m, q, new_model_path),
                             lambda n, m, q: TestLinear(n, m, q, new_model_path), "ReconTest")
