"""Optional visdom line plots -- the part of the reference's `vis_tools.py::visualizer`
(vis_tools.py:20-57) that train_gan.py uses.  Imported only when visdom is installed and a
server is reachable; training never depends on it."""
import numpy as np


class visualizer(object):
    def __init__(self, port=8000, scatter_size=[[-1, 1], [-1, 1]], env_name="main"):
        from visdom import Visdom
        self.vis = Visdom(port=port)
        self.env = env_name
        self.plots = {}

    def plot(self, var_name, split_name, title_name, x, y):
        if var_name not in self.plots:
            self.plots[var_name] = self.vis.line(
                X=np.array([x, x]), Y=np.array([y, y]), env=self.env,
                opts=dict(legend=[split_name], title=title_name, xlabel="Epochs", ylabel=var_name))
        else:
            self.vis.line(X=np.array([x]), Y=np.array([y]), env=self.env, win=self.plots[var_name],
                          name=split_name, update="append")
