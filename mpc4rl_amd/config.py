"""YAML problem loader compatible with the reference's ``config/cartpole.yaml`` keys (rlmpc/common/utils.py:41-47,
rlmpc/mpc/cartpole/acados.py:111-186) and iterate store / load (rlmpc/examples/chain_mass.py:119-120)."""
from __future__ import annotations

import numpy as np
import torch
import yaml

from .problems import OcpDescription, cartpole_ocp


def read_config(config_file: str) -> dict:
    """rlmpc/common/utils.py:41-47."""
    with open(config_file, "r") as stream:
        return yaml.safe_load(stream)


def cartpole_ocp_from_config(config: dict) -> OcpDescription:
    """``config`` = the ``mpc`` block of config/cartpole.yaml (or the whole file).  Keys used: ocp_options.{tf,
    sim_method_num_stages, nlp_solver_max_iter}, dimensions.N, cost.{W_0, W, W_e, yref_0, yref, yref_e}, model.params.{M,m,l}.value,
    constraints.{lbu, ubu, lbx, ubx, lbx_e, ubx_e, x0}."""
    c = config.get("mpc", config)
    opt, dims, cost, cons = c["ocp_options"], c["dimensions"], c["cost"], c["constraints"]
    prm = c["model"]["params"]
    if prm["g"]["value"] != 9.8 or not prm["g"].get("fixed", True):
        raise ValueError("g is a compiled-in constant (9.8, fixed) of the cartpole model")
    ocp = cartpole_ocp(N=dims["N"], tf=opt["tf"], M=prm["M"]["value"], m=prm["m"]["value"], l=prm["l"]["value"],
                       W=np.array(cost["W"]), W_e=np.array(cost["W_e"]), yref=np.array(cost["yref"]),
                       yref_e=np.array(cost["yref_e"]), W_0=np.array(cost["W_0"]) if "W_0" in cost else None,
                       yref_0=np.array(cost["yref_0"]) if "yref_0" in cost else None, max_iter=opt.get("nlp_solver_max_iter", 500))
    stages = opt.get("sim_method_num_stages", 4)
    ocp.h = opt["tf"] / dims["N"] / stages                      # cartpole/acados.py:86-92
    ocp.lbu, ocp.ubu = np.array(cons["lbu"], float), np.array(cons["ubu"], float)
    ocp.idxbx, ocp.lbx, ocp.ubx = np.array(cons["idxbx"]), np.array(cons["lbx"], float), np.array(cons["ubx"], float)
    ocp.idxbx_e, ocp.lbx_e, ocp.ubx_e = np.array(cons["idxbx_e"]), np.array(cons["lbx_e"], float), np.array(cons["ubx_e"], float)
    ocp.x0 = np.array(cons["x0"], float)
    return ocp


def store_iterate(mpc_batch, filename: str) -> None:
    x, u, pi, bnd, res = mpc_batch.get_iterate()
    torch.save({"x": x.cpu(), "u": u.cpu(), "pi": pi.cpu(), "bnd": bnd.cpu()}, filename)


def load_iterate(mpc_batch, filename: str) -> None:
    d = torch.load(filename)
    mpc_batch.set_iterate(d["x"], d["u"], d["pi"], d["bnd"])
