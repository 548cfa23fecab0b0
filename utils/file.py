from ndivplanning_amd.utils.file import (AttrDict, DotMap, load_training_config_file,  # noqa: F401
                                         make_paths_absolute)
