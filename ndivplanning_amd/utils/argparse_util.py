"""argparse helpers -- mirror of the reference's `utils/argparse_util.py`: path-checking
`type=` callables (argparse.ArgumentTypeError on failure, argparse_util.py:26-59) and the
CLI-over-YAML merge (argparse_util.py:62-71)."""
import argparse
import glob
import os


def listdir_nohidden(path):
    """Entries of `path` whose names do not start with a dot (glob skips dot-files)."""
    return glob.glob(os.path.join(path, "*"))


def file_exists(prospective_file):
    """argparse `type=`: resolve against the working directory, reject a missing file
    with the reference's message (argparse_util.py:26-31)."""
    resolved = os.path.join(os.getcwd(), prospective_file)
    if os.path.exists(resolved):
        return resolved
    raise argparse.ArgumentTypeError("File: '{0}' does not exist".format(resolved))


def _checked_dir(prospective_dir, mode, word):
    dir_path = os.path.join(os.getcwd(), prospective_dir)
    if not os.path.isdir(dir_path):
        raise argparse.ArgumentTypeError("Directory: '{0}' does not exist".format(dir_path))
    if not os.access(dir_path, mode):
        raise argparse.ArgumentTypeError("Directory: '{0}' is not {1}".format(dir_path, word))
    return dir_path


def dir_exists_write_privileges(prospective_dir):
    return _checked_dir(prospective_dir, os.W_OK, "writable")


def dir_exists_read_privileges(prospective_dir):
    return _checked_dir(prospective_dir, os.R_OK, "readable")


def override_dotmap(namespace, config_key):
    """Every CLI argument that was given becomes a TOP-LEVEL key of the YAML config
    (reference argparse_util.py:62-71)."""
    cfg = getattr(namespace, config_key)
    for arg in vars(namespace):
        if arg != config_key and getattr(namespace, arg) is not None:
            cfg[arg] = getattr(namespace, arg)
    return cfg
