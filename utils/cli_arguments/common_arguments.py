from ndivplanning_amd.utils.cli_arguments.common_arguments import add_common_arguments  # noqa: F401
