from .preprocessing import (SceneTable, collate, get_node_timestep_data, get_timesteps_data, present_nodes,  # noqa: F401
                            restore)
