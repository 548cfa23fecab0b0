from ndivplanning_amd.utils.trajectory_loader import PushDataset, SyntheticPushDataset  # noqa: F401
