from .preprocessing import SceneTable, get_timesteps_data, get_node_timestep_data, present_nodes  # noqa: F401
