"""The shared CLI of the training scripts -- mirror of the reference's
`utils/cli_arguments/common_arguments.py:7-62`: same flags, same types, same help."""
from argparse import ArgumentParser

from ..argparse_util import (dir_exists_read_privileges, dir_exists_write_privileges, file_exists)
from ..file import load_training_config_file

_FLAGS = [
    ("--config-file", dict(type=load_training_config_file, default="config/default.yaml",
                           help="Config file absolute path. CLI takes priority over config file")),
    ("--log-port", dict(type=int, help="Port number of logging server")),
    ("--gpu-id", dict(type=int, help="GPU id for single gpu training")),
    ("--trajectory-length", dict(type=int, help="Trajectory length to use for training")),
    ("--log-dir", dict(type=dir_exists_write_privileges, help="Log file storage directory path")),
    ("--forward-save-path", dict(type=dir_exists_write_privileges, help="Forward model storage directory path")),
    ("--gan-save-path", dict(type=dir_exists_write_privileges, help="GAN model storage directory path")),
    ("--train-data-path", dict(type=dir_exists_read_privileges, help="Train data file storage directory path")),
    ("--evaluation-data-path", dict(type=dir_exists_read_privileges,
                                    help="Evaluation data file storage directory path")),
    ("--restore-weights", dict(type=file_exists, help="Restore model with pre-trained weights")),
]


def add_common_arguments(parser: ArgumentParser) -> ArgumentParser:
    core = parser.add_argument_group("core arguments")
    for flag, kw in _FLAGS:
        core.add_argument(flag, **kw)
    return parser
