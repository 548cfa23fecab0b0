"""Config loading -- mirror of the reference's `utils/file.py` (same function names and
behaviour: YAML -> attribute dictionary, `*_path` keys made absolute).

Differences forced by this environment, none visible to a config author:
  * `dotmap` is not installed: `AttrDict` below provides the part of DotMap the training
    scripts use (attribute access, nested dicts, missing key -> empty AttrDict);
  * `yaml.load(stream)` without a Loader (utils/file.py:26) raises on PyYAML >= 6;
    `yaml.safe_load` reads the same files.
"""
import logging
import os
import sys

import yaml

logging.basicConfig(format="%(asctime)s %(levelname)s %(message)s", level=logging.DEBUG, stream=sys.stdout)


class AttrDict(dict):
    """dict with attribute access; nested dicts convert on the way in; a missing key reads
    as an empty AttrDict (DotMap's behaviour, which the reference relies on)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        super().__setitem__(key, value)

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        if key not in self:
            self[key] = AttrDict()
        return self[key]

    def __setattr__(self, key, value):
        self[key] = value

    def toDict(self):
        return {k: (v.toDict() if isinstance(v, AttrDict) else v) for k, v in self.items()}


DotMap = AttrDict   # the name the reference's scripts import


def load_training_config_file(filename):
    """Load a training configuration yaml file into an attribute dictionary
    (reference utils/file.py:20-30; used as the argparse `type=` of --config-file)."""
    print("Loading training configuration file: {0}".format(filename))
    config_file_path = os.path.join(os.getcwd(), filename)
    with open(config_file_path, "r") as stream:
        cfg = AttrDict(yaml.safe_load(stream) or {})
    return make_paths_absolute(os.getcwd(), cfg)


def make_paths_absolute(dir_, cfg, log_not_exist=False):
    """Make every value whose key ends in `_path` absolute w.r.t. dir_ (utils/file.py:33-54).
    Values with a scheme prefix (`synthetic:...`) are not filesystem paths and stay as they are."""
    for key in list(cfg.keys()):
        val = cfg[key]
        if key.endswith("_path") and isinstance(val, str) and ":" not in val and dir_ not in val:
            cfg[key] = os.path.abspath(os.path.join(dir_, val))
            if not os.path.isfile(cfg[key]) and log_not_exist:
                logging.error("%s does not exist.", cfg[key])
        if isinstance(val, AttrDict):
            cfg[key] = make_paths_absolute(dir_, val)
    return cfg
