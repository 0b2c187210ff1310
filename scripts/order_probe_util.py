import numpy as np
from gfasort_amd import graph as G

def permuted(g, perm):
    """relabel dense indices: new index perm[k] for old index k"""
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    return G.FlatGraph(node_len=g.node_len[inv], step_node=perm[g.step_node].astype(np.uint32), step_is_rev=g.step_is_rev,
                       path_first_step=g.path_first_step, node_ids=g.node_ids[inv], path_names=g.path_names, step_node_id=g.step_node_id)
