"""BertConfig: the hyper-parameter bag of the two-stream model (drop-in for the reference's
`models/vilbert_dialog.py:131-274`).

Same surface: `BertConfig(vocab_size_int | json_path, **overrides)`, `from_dict`, `from_json_file`,
`to_dict`, `to_json_string`; `from_json_file` = constructor defaults overwritten by the JSON keys
(:257-262), so `fusion_method`, `with_coattention`, `fast_mode`, ... keep their defaults with
`config/bert_base_6layer_6conect.json`."""
from __future__ import annotations

import copy
import json

_DEFAULTS = dict(
    hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
    hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
    max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02,
    v_feature_size=2048, v_target_size=1601, v_hidden_size=768, v_num_hidden_layers=3,
    v_num_attention_heads=12, v_intermediate_size=3072, bi_hidden_size=1024,
    bi_num_attention_heads=16, v_attention_probs_dropout_prob=0.1, v_hidden_act="gelu",
    v_hidden_dropout_prob=0.1, v_initializer_range=0.2, v_biattention_id=[0, 1],
    t_biattention_id=[10, 11], predict_feature=False, fast_mode=False, fixed_v_layer=0,
    fixed_t_layer=0, in_batch_pairs=False, fusion_method="mul", intra_gate=False,
    with_coattention=True,
)


class BertConfig(object):
    def __init__(self, vocab_size_or_config_json_file, **kwargs):
        unknown = set(kwargs) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"unexpected BertConfig arguments: {sorted(unknown)}")
        values = dict(_DEFAULTS)
        values.update(kwargs)
        # same sanity checks as the reference constructor (:196-198)
        assert len(values["v_biattention_id"]) == len(values["t_biattention_id"])
        assert max(values["v_biattention_id"]) < values["v_num_hidden_layers"]
        assert max(values["t_biattention_id"]) < values["num_hidden_layers"]
        if isinstance(vocab_size_or_config_json_file, str):
            with open(vocab_size_or_config_json_file, "r", encoding="utf-8") as reader:
                for key, value in json.loads(reader.read()).items():
                    self.__dict__[key] = value
        elif isinstance(vocab_size_or_config_json_file, int) and not isinstance(vocab_size_or_config_json_file, bool):
            self.vocab_size = vocab_size_or_config_json_file
            for key, value in values.items():
                setattr(self, key, copy.deepcopy(value))
        else:
            raise ValueError("First argument must be either a vocabulary size (int)"
                             "or the path to a pretrained model config file (str)")

    @classmethod
    def from_dict(cls, json_object):
        config = cls(vocab_size_or_config_json_file=-1)
        for key, value in json_object.items():
            config.__dict__[key] = value
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as reader:
            return cls.from_dict(json.loads(reader.read()))

    def __repr__(self):
        return str(self.to_json_string())

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def validate_for_hip(self):
        """Shapes the gfx950 kernels are built for; raised early with a clear message."""
        def need(cond, msg):
            if not cond:
                raise ValueError("unimm_amd (HIP path): " + msg)
        dt = self.hidden_size // self.num_attention_heads
        dv = self.v_hidden_size // self.v_num_attention_heads
        db = self.bi_hidden_size // self.bi_num_attention_heads
        need(self.hidden_size % self.num_attention_heads == 0 and dt in (64, 128), f"text head size {dt} not in (64,128)")
        need(self.v_hidden_size % self.v_num_attention_heads == 0 and dv in (64, 128), f"image head size {dv} not in (64,128)")
        need(self.bi_hidden_size % self.bi_num_attention_heads == 0 and db in (64, 128), f"bi head size {db} not in (64,128)")
        for n in ("hidden_size", "v_hidden_size", "bi_hidden_size", "intermediate_size", "v_intermediate_size",
                  "v_feature_size"):
            need(getattr(self, n) % 64 == 0, f"{n} must be a multiple of 64")
        need(max(self.hidden_size, self.v_hidden_size) <= 1024, "hidden sizes above 1024 not supported by the row kernels")
        need(self.type_vocab_size == 2, "type_vocab_size must be 2")
        need(self.hidden_act == "gelu" and self.v_hidden_act == "gelu", "only erf-GELU is implemented")
        need(self.fusion_method in ("mul", "sum"), "fusion_method must be 'mul' or 'sum' (models/vilbert_dialog.py:1062-1067 asserts on anything else)")
        need(not (self.fast_mode or self.in_batch_pairs), "fast_mode / in_batch_pairs (every text paired with every image of the batch, "
             "models/vilbert_dialog.py:876-899) are not built: off in bert_base_6layer_6conect.json and unused by the reference's scripts")
        for v_end, t_end in zip(self.v_biattention_id, self.t_biattention_id):       # the reference's own asserts (:847-848)
            assert self.fixed_t_layer <= t_end
            assert self.fixed_v_layer <= v_end
