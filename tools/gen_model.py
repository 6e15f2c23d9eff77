"""Model compiler: cassie.xml (MJCF, read from the reference tree in THIS container) -> numeric constant tables.

Emits (derived data, not a copy of the XML):
  apex_amd/cassie_model.json        everything below, for Python-side consumers and tests
  oracle/cassie_model_gen.h         `static const double` tables for the fp64 CPU oracle
  apex_amd/csrc/cassie_model_gen.h  `float` tables for the HIP kernels

Follows MuJoCo's documented compile rules for the subset cassie.xml uses (cassie/cassiemujoco/cassie.xml:3-268):
angles in degrees, xyaxes -> quaternion, fromto -> capsule centre/axis/half-length, explicit <inertial>,
connect anchors stored in both body frames (evaluated at qpos0), joint `ref` -> qpos0, defaults classes for geoms.
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, ".."))
XML = "/root/reference/cassie/cassiemujoco/cassie.xml"

# init pose baked into libcassiemujoco.so's .rodata (read with struct.unpack; data, not code)
INIT_QPOS = [0, 0, 1.01, 1, 0, 0, 0,
             0.0045, 0, 0.4973, 0.9785, -0.0164, 0.01787, -0.2049, -1.1997, 0, 1.4267, 0, -1.5244, 1.5244, -1.5968,
             -0.0045, 0, 0.4973, 0.9786, 0.00386, -0.01524, -0.2051, -1.1997, 0, 1.4267, 0, -1.5244, 1.5244, -1.5968]


def fl(s):
    return [float(x) for x in s.split()]


def mat2quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    q /= np.linalg.norm(q)
    return q if q[0] >= 0 else -q


def quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def xyaxes2quat(v):
    x = np.array(v[:3]); y = np.array(v[3:])
    x = x / np.linalg.norm(x)
    y = y - x * (x @ y); y /= np.linalg.norm(y)
    z = np.cross(x, y)
    return mat2quat(np.stack([x, y, z], axis=1))


def compile_model(xml_path=XML):
    root = ET.parse(xml_path).getroot()
    opt = root.find("option").attrib
    # geom default classes (cassie.xml:14-36): only contype/conaffinity/condim matter for collision primitives
    classes = {"collision": dict(contype=1, conaffinity=0), "collision-left": dict(contype=2, conaffinity=4),
               "collision-right": dict(contype=4, conaffinity=2)}
    bodies, joints, dofs, geoms = [], [], [], []
    name2body = {"world": 0}
    bodies.append(dict(name="world", parent=0, pos=[0, 0, 0], quat=[1, 0, 0, 0], ipos=[0, 0, 0], mass=0.0,
                       inertia=np.zeros((3, 3)).tolist(), dofadr=-1, dofnum=0))
    nq = [0]

    def walk(elem, parent):
        bid = len(bodies)
        a = elem.attrib
        quat = xyaxes2quat(fl(a["xyaxes"])) if "xyaxes" in a else np.array([1.0, 0, 0, 0])
        ine = elem.find("inertial").attrib
        fi = fl(ine["fullinertia"])     # xx yy zz xy xz yz, about the COM, body-frame aligned (no inertial quat)
        I = np.array([[fi[0], fi[3], fi[4]], [fi[3], fi[1], fi[5]], [fi[4], fi[5], fi[2]]])
        b = dict(name=a["name"], parent=parent, pos=fl(a.get("pos", "0 0 0")), quat=quat.tolist(), ipos=fl(ine["pos"]),
                 mass=float(ine["mass"]), inertia=I.tolist(), dofadr=len(dofs), dofnum=0)
        bodies.append(b)
        name2body[a["name"]] = bid
        for j in elem.findall("joint"):
            ja = j.attrib
            jt = ja.get("type", "hinge")
            limited = ja.get("limited", "true") == "true"      # <default><joint limited='true'/>
            rng = np.deg2rad(fl(ja["range"])).tolist() if "range" in ja else [0.0, 0.0]
            if jt in ("ball",):
                ref = 0.0
            else:
                ref = float(ja.get("ref", 0.0))
                if jt == "hinge":
                    ref = float(np.deg2rad(ref))
            jd = dict(name=ja.get("name", a["name"] + "-" + jt), type=jt, body=bid, qposadr=nq[0], dofadr=len(dofs),
                      axis=fl(ja.get("axis", "0 0 1")), ref=ref, limited=bool(limited and jt in ("hinge", "slide")),
                      range=rng, stiffness=float(ja.get("stiffness", 0)), damping=float(ja.get("damping", 0)),
                      armature=float(ja.get("armature", 0)))
            joints.append(jd)
            nd = 3 if jt == "ball" else 1
            for k in range(nd):
                dofs.append(dict(joint=len(joints) - 1, body=bid, type=jt, sub=k, damping=jd["damping"],
                                 armature=jd["armature"], stiffness=jd["stiffness"] if jt != "ball" else 0.0))
            nq[0] += 4 if jt == "ball" else 1
            b["dofnum"] += nd
        for gm in elem.findall("geom"):
            ga = gm.attrib
            if ga.get("class") not in classes:
                continue                          # visual meshes: contype 0 / conaffinity 0
            cl = classes[ga["class"]]
            if ga["type"] == "sphere":
                geoms.append(dict(body=bid, type="sphere", radius=fl(ga["size"])[0], pos=fl(ga["pos"]), axis=[0, 0, 1],
                                  half=0.0, **cl))
            elif ga["type"] == "capsule":
                ft = np.array(fl(ga["fromto"]))
                p0, p1 = ft[:3], ft[3:]
                d = p1 - p0
                geoms.append(dict(body=bid, type="capsule", radius=fl(ga["size"])[0], pos=((p0 + p1) / 2).tolist(),
                                  axis=(d / np.linalg.norm(d)).tolist(), half=float(np.linalg.norm(d) / 2), **cl))
        for child in elem.findall("body"):
            walk(child, bid)

    wb = root.find("worldbody")
    floor = [g for g in wb.findall("geom") if g.attrib.get("name") == "floor"][0].attrib
    for child in wb.findall("body"):
        walk(child, 0)

    # contact priority order used by BOTH the oracle and the kernel when the per-step contact cap binds:
    # feet, tarsi, shins, hip-pitch capsules, pelvis sphere (MuJoCo itself orders by geom id)
    def _prio(g):
        nm = bodies[g["body"]]["name"]
        for k, key in enumerate(["foot", "tarsus", "shin", "hip-pitch", "pelvis"]):
            if nm.endswith(key):
                return (k, 0 if nm.startswith("left") else 1)
        raise ValueError(nm)
    geoms.sort(key=_prio)

    nbody, nv = len(bodies), len(dofs)
    assert nbody == 26 and nv == 32 and nq[0] == 35, (nbody, nv, nq[0])

    # qpos0: hinge/slide ref, ball identity
    qpos0 = np.zeros(35)
    for j in joints:
        if j["type"] == "ball":
            qpos0[j["qposadr"]] = 1.0
        else:
            qpos0[j["qposadr"]] = j["ref"]

    # forward kinematics at qpos0 == the configuration the XML defines (all joint displacements zero)
    xpos = [np.zeros(3)] * nbody
    xmat = [np.eye(3)] * nbody
    for i in range(1, nbody):
        p = bodies[i]["parent"]
        xpos[i] = xpos[p] + xmat[p] @ np.array(bodies[i]["pos"])
        xmat[i] = xmat[p] @ quat2mat(bodies[i]["quat"])

    eqs = []
    for c in root.find("equality").findall("connect"):
        b1, b2 = name2body[c.attrib["body1"]], name2body[c.attrib["body2"]]
        a1 = np.array(fl(c.attrib["anchor"]))
        world = xpos[b1] + xmat[b1] @ a1
        a2 = xmat[b2].T @ (world - xpos[b2])
        eqs.append(dict(body1=b1, body2=b2, anchor1=a1.tolist(), anchor2=a2.tolist()))

    name2joint = {j["name"]: k for k, j in enumerate(joints)}
    acts = []
    for m in root.find("actuator").findall("motor"):
        ma = m.attrib
        j = joints[name2joint[ma["joint"]]]
        acts.append(dict(name=ma["name"], dof=j["dofadr"], qposadr=j["qposadr"], gear=float(ma["gear"]),
                         ctrlmax=fl(ma["ctrlrange"])[1], rpm=float(ma["user"])))
    sens = root.find("sensor")
    motor_bits = [int(s.attrib["user"]) for s in sens.findall("actuatorpos")]
    jsens = []
    for s in sens.findall("jointpos"):
        j = joints[name2joint[s.attrib["joint"]]]
        jsens.append(dict(name=s.attrib["name"], qposadr=j["qposadr"], dofadr=j["dofadr"], bits=int(s.attrib["user"])))
    imu = [s for s in root.iter("site") if s.attrib.get("name") == "imu"][0]

    model = dict(
        timestep=float(opt["timestep"]), iterations=int(opt["iterations"]), gravity=fl(opt["gravity"]),
        nq=35, nv=nv, nbody=nbody, bodies=bodies, joints=joints, dofs=dofs, geoms=geoms, equalities=eqs,
        actuators=acts, motor_bits=motor_bits, joint_sensors=jsens, imu_pos=fl(imu.attrib["pos"]),
        floor_pos=fl(floor["pos"]), qpos0=qpos0.tolist(), init_qpos=INIT_QPOS,
        # MuJoCo 2.0 documented defaults for everything cassie.xml leaves unset
        solref=[0.005, 1.0], solimp=[0.9, 0.95, 0.001, 0.5, 2.0], limit_solref=[0.02, 1.0],
        friction_default=[1.0, 0.005, 0.0001], impratio=1.0,
    )
    return model


DECL = "static const"


def _carr(name, arr, ctype, per_line=8):
    flat = np.asarray(arr).reshape(-1)
    isint = ctype == "int"
    body = []
    for i in range(0, len(flat), per_line):
        chunk = flat[i:i + per_line]
        body.append("    " + ", ".join((str(int(v)) if isint else (repr(float(v)) + ("f" if ctype == "float" else "")))
                                       for v in chunk))
    return f"{DECL} {ctype} {name}[{len(flat)}] = {{\n" + ",\n".join(body) + "\n};\n"


def emit_header(model, ctype, guard, decl="static const"):
    global DECL
    DECL = decl
    m = model
    B, D, J = m["bodies"], m["dofs"], m["joints"]
    jt = {"slide": 0, "hinge": 1, "ball": 2}
    out = [f"// GENERATED by tools/gen_model.py from the reference's cassie.xml (numeric tables only) — do not edit.\n"
           f"#ifndef {guard}\n#define {guard}\n",
           f"#define CM_NBODY {m['nbody']}\n#define CM_NV {m['nv']}\n#define CM_NQ {m['nq']}\n"
           f"#define CM_NJNT {len(J)}\n#define CM_NGEOM {len(m['geoms'])}\n#define CM_NEQ {len(m['equalities'])}\n"
           f"#define CM_NU {len(m['actuators'])}\n#define CM_NJSENS {len(m['joint_sensors'])}\n"]
    R = ctype
    out.append(_carr("cm_body_parent", [b["parent"] for b in B], "int"))
    out.append(_carr("cm_body_dofadr", [b["dofadr"] for b in B], "int"))
    out.append(_carr("cm_body_dofnum", [b["dofnum"] for b in B], "int"))
    out.append(_carr("cm_body_pos", [b["pos"] for b in B], R, 3))
    out.append(_carr("cm_body_quat", [b["quat"] for b in B], R, 4))
    out.append(_carr("cm_body_ipos", [b["ipos"] for b in B], R, 3))
    out.append(_carr("cm_body_mass", [b["mass"] for b in B], R))
    out.append(_carr("cm_body_inertia", [b["inertia"] for b in B], R, 9))
    out.append(_carr("cm_jnt_type", [jt[j["type"]] for j in J], "int"))
    out.append(_carr("cm_jnt_body", [j["body"] for j in J], "int"))
    out.append(_carr("cm_jnt_qposadr", [j["qposadr"] for j in J], "int"))
    out.append(_carr("cm_jnt_dofadr", [j["dofadr"] for j in J], "int"))
    out.append(_carr("cm_jnt_axis", [j["axis"] for j in J], R, 3))
    out.append(_carr("cm_jnt_ref", [j["ref"] for j in J], R))
    out.append(_carr("cm_jnt_limited", [int(j["limited"]) for j in J], "int"))
    out.append(_carr("cm_jnt_range", [j["range"] for j in J], R, 2))
    out.append(_carr("cm_jnt_stiffness", [j["stiffness"] for j in J], R))
    out.append(_carr("cm_dof_jnt", [d["joint"] for d in D], "int"))
    out.append(_carr("cm_dof_body", [d["body"] for d in D], "int"))
    # dof parent chain (MuJoCo dof_parentid): previous dof in the same body, else last dof of the nearest ancestor with dofs
    par = []
    for i, d in enumerate(D):
        b = d["body"]
        if i > B[b]["dofadr"]:
            par.append(i - 1)
        else:
            p = B[b]["parent"]
            while p > 0 and B[p]["dofnum"] == 0:
                p = B[p]["parent"]
            par.append(B[p]["dofadr"] + B[p]["dofnum"] - 1 if p > 0 else -1)
    out.append(_carr("cm_dof_parent", par, "int"))
    # sparse mass-matrix addressing (MuJoCo dof_Madr): row i holds M[i][i], M[i][parent(i)], ... up to the root
    madr, tot = [], 0
    for i in range(len(D)):
        madr.append(tot)
        j = i
        while j >= 0:
            tot += 1
            j = par[j]
    out.append(_carr("cm_dof_madr", madr + [tot], "int"))
    out.append(f"#define CM_NM {tot}\n")
    # ancestor chains (self first), padded to 16, and chain lengths: compile-time unrolling tables
    depth, anc = [], []
    for i in range(len(D)):
        ch, j = [], i
        while j >= 0:
            ch.append(j); j = par[j]
        depth.append(len(ch)); anc.append(ch + [-1] * (16 - len(ch)))
    out.append(_carr("cm_dof_depth", depth, "int"))
    out.append(_carr("cm_dof_anc", anc, "int", 16))
    jadr, jnum = [], []
    for bi in range(len(B)):
        js = [k for k, j in enumerate(J) if j["body"] == bi]
        jadr.append(js[0] if js else -1); jnum.append(len(js))
    out.append(_carr("cm_body_jntadr", jadr, "int"))
    out.append(_carr("cm_body_jntnum", jnum, "int"))
    lastdof = []
    for bi in range(len(B)):
        b = bi
        while b > 0 and B[b]["dofnum"] == 0:
            b = B[b]["parent"]
        lastdof.append(B[b]["dofadr"] + B[b]["dofnum"] - 1 if b > 0 else -1)
    out.append(_carr("cm_body_lastdof", lastdof, "int"))
    kids = [[k for k in range(len(B)) if k > 0 and B[k]["parent"] == bi and k != bi] for bi in range(len(B))]
    out.append(_carr("cm_body_nchild", [len(k) for k in kids], "int"))
    out.append(_carr("cm_body_child", [k + [-1] * (4 - len(k)) for k in kids], "int", 4))
    out.append(_carr("cm_dof_damping", [d["damping"] for d in D], R))
    out.append(_carr("cm_dof_armature", [d["armature"] for d in D], R))
    G = m["geoms"]
    out.append(_carr("cm_geom_body", [g["body"] for g in G], "int"))
    out.append(_carr("cm_geom_iscapsule", [int(g["type"] == "capsule") for g in G], "int"))
    out.append(_carr("cm_geom_contype", [g["contype"] for g in G], "int"))
    out.append(_carr("cm_geom_conaffinity", [g["conaffinity"] for g in G], "int"))
    out.append(_carr("cm_geom_radius", [g["radius"] for g in G], R))
    out.append(_carr("cm_geom_half", [g["half"] for g in G], R))
    out.append(_carr("cm_geom_pos", [g["pos"] for g in G], R, 3))
    out.append(_carr("cm_geom_axis", [g["axis"] for g in G], R, 3))
    E = m["equalities"]
    out.append(_carr("cm_eq_body1", [e["body1"] for e in E], "int"))
    out.append(_carr("cm_eq_body2", [e["body2"] for e in E], "int"))
    out.append(_carr("cm_eq_anchor1", [e["anchor1"] for e in E], R, 3))
    out.append(_carr("cm_eq_anchor2", [e["anchor2"] for e in E], R, 3))
    A = m["actuators"]
    out.append(_carr("cm_act_dof", [a["dof"] for a in A], "int"))
    out.append(_carr("cm_act_qposadr", [a["qposadr"] for a in A], "int"))
    out.append(_carr("cm_act_gear", [a["gear"] for a in A], R))
    out.append(_carr("cm_act_ctrlmax", [a["ctrlmax"] for a in A], R))
    out.append(_carr("cm_act_rpm", [a["rpm"] for a in A], R))
    out.append(_carr("cm_act_bits", m["motor_bits"], "int"))
    S = m["joint_sensors"]
    out.append(_carr("cm_jsens_qposadr", [s["qposadr"] for s in S], "int"))
    out.append(_carr("cm_jsens_dofadr", [s["dofadr"] for s in S], "int"))
    out.append(_carr("cm_jsens_bits", [s["bits"] for s in S], "int"))
    out.append(_carr("cm_imu_pos", m["imu_pos"], R, 3))
    out.append(_carr("cm_floor_pos", m["floor_pos"], R, 3))
    out.append(_carr("cm_qpos0", m["qpos0"], R))
    out.append(_carr("cm_init_qpos", m["init_qpos"], R))
    out.append(f"#endif  // {guard}\n")
    return "\n".join(out)


def main():
    model = compile_model()
    with open(os.path.join(REPO, "apex_amd", "cassie_model.json"), "w") as f:
        json.dump(model, f, indent=1)
    with open(os.path.join(REPO, "oracle", "cassie_model_gen.h"), "w") as f:
        f.write(emit_header(model, "double", "ORACLE_CASSIE_MODEL_GEN_H"))
    with open(os.path.join(REPO, "apex_amd", "csrc", "cassie_model_gen.h"), "w") as f:
        f.write(emit_header(model, "float", "APX_CASSIE_MODEL_GEN_H", decl="static __device__ const"))
    with open(os.path.join(REPO, "apex_amd", "csrc", "cassie_tables.h"), "w") as f:
        txt = emit_header(model, "float", "APX_CASSIE_TABLES_H", decl="constexpr")
        txt = txt.replace("cm_", "ct_")      # constexpr twins of the device tables: distinct names, usable as constants
        txt = txt.replace("#define APX_CASSIE_TABLES_H\n", "#define APX_CASSIE_TABLES_H\nnamespace cmt {\n").replace("#endif  // APX_CASSIE_TABLES_H", "}  // namespace cmt\n#endif  // APX_CASSIE_TABLES_H")
        f.write(txt)
    tot = sum(b["mass"] for b in model["bodies"])
    print(f"nbody={model['nbody']} nv={model['nv']} ngeom(collision)={len(model['geoms'])} total mass={tot:.3f}")
    for e in model["equalities"]:
        print("connect", e)
    for i, j in enumerate(model["joints"]):
        print(i, j["name"], j["type"], "qadr", j["qposadr"], "dadr", j["dofadr"], "ref", round(j["ref"], 4), "lim", j["limited"])


if __name__ == "__main__":
    main()
