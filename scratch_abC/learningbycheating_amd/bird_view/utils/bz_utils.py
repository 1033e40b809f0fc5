"""Logging + the config.json half of the checkpoint contract (reference bird_view/utils/bz_utils/saver.py:51-136).
`benchmark_agent.py:48` reads <log_dir>/config.json next to model-N.th, so save_config keeps the reference's schema
and its stringification rule (non int/float leaves become str, saver.py:82-95).  loguru/tensorboardX are replaced by
a JSONL scalar log."""
import copy
import json
import time
from collections import OrderedDict
from pathlib import Path

import numpy as np


def load_json(path):
    with open(path, "r") as f:
        return json.load(f)


class Experiment(object):
    def init(self, log_dir, rank=0):
        self.epoch = 0
        self.rank = rank
        self.scalars = OrderedDict()
        self.log_dir = Path(log_dir).resolve()
        if rank == 0:
            self.log_dir.mkdir(parents=True, exist_ok=True)
        return self

    def load_config(self, model_path):
        return load_json(str(Path(model_path).parent / "config.json"))

    def save_config(self, config_dict):
        def _process(x):
            for key, val in x.items():
                if isinstance(val, dict):
                    _process(val)
                elif not isinstance(val, float) and not isinstance(val, int):
                    x[key] = str(val)
        config = copy.deepcopy(config_dict)
        _process(config)
        if self.rank == 0:
            with open(str(self.log_dir / "config.json"), "w+") as f:
                json.dump(config, f, indent=4, sort_keys=True)

    def scalar(self, is_train=True, **kwargs):
        for k, v in sorted(kwargs.items()):
            self.scalars.setdefault((is_train, k), []).append(float(v))

    def image(self, is_train=True, **kwargs):
        pass   # visualisation (cv2) is outside the hot path

    def end_epoch(self, net=None):
        rec = {"epoch": self.epoch, "time": time.time()}
        for (is_train, k), v in self.scalars.items():
            name = "%s_%s" % ("train" if is_train else "val", k)
            rec[name] = {"mean": float(np.mean(v)), "std": float(np.std(v)), "min": float(np.min(v)), "max": float(np.max(v)), "n": len(v)}
        if self.rank == 0:
            with open(str(self.log_dir / "log.jsonl"), "a") as f:
                f.write(json.dumps(rec) + "\n")
        self.scalars.clear()
        self.epoch += 1
        return rec


log = Experiment()
