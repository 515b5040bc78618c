"""A plan as a DAG: independent branches of a compiled / lifted plan on different lanes (HIP streams) of one context, so that a recorded
hipGraph has parallel branches instead of one chain (include/lele_hip.h, lele_hip_lane_*).

lele's generated forward() is a sequence of kernel calls (src/compiler/mod.rs:1291-1303) because a CPU thread runs one kernel at a time.
On the device the same sequence leaves the chip idle wherever a kernel's grid is small: the FSMN memory block of a SenseVoice layer
(11 us) waits behind the attention (25 us) it does not depend on, the three detection-head scales of Yolo26n-seg
(examples/yolo26n-seg/src/yolo26seg.rs:509-627) behind each other.  The plan knows every dependency; `schedule` turns them into lanes:

  1. dependencies: a statement depends on the producers of the values it reads -- through views (reshape / flatten / chview share their
     source's buffer) and channel windows (a Concat written in place has one writer per operand: a reader of the whole depends on all).
  2. list scheduling in plan order over `lanes` streams with measured per-statement device times (Runner.stmt_times): a statement goes
     to the lane where it can start first; leaving the lane of its latest operand has to buy at least `min_gain_ms`.
  3. ordering: every cross-lane dependency becomes record (after the producer) / wait (before the consumer) on an event; vector clocks
     drop the waits a lane already has transitively.
  4. buffers: the sequential plan re-uses a slot as soon as its value is dead IN PLAN ORDER, which says nothing once two lanes run
     side by side.  Slots are re-assigned under happens-before: a slot may take a new value only if EVERY access of its previous value
     (its writers and all readers, through every alias) happens-before the new writer by the vector clocks -- correct for any timing,
     not just the measured one.

The result is a plan of the same format with "lane" / "wait" / "record" on its device statements (+ a final "join"); lele_amd.plan.Runner
issues it on lanes, eagerly or under graph capture.  Same kernels, same arguments, same arithmetic: outputs are bit-identical to the
sequential plan's (tests/test_lanes.py)."""
import copy


def _refs(n, acc):
    if isinstance(n, dict):
        for key in ("ref", "ints"):
            if isinstance(n.get(key), str):
                acc.append(n[key])
        if isinstance(n.get("refs"), list):
            acc += n["refs"]
        for v in n.values():
            _refs(v, acc)
    elif isinstance(n, list):
        for v in n:
            _refs(v, acc)
    return acc


_VIEW_FNS = ("reshape", "flatten", "unsqueeze", "squeeze", "identity")
_HOST_OPS = ("ints", "host")


def schedule(plan, times=None, lanes=3, min_gain_ms=0.004, default_ms=0.006):
    """-> a new plan with lanes, or None when the plan has a construct this pass does not order (an `If`, a host statement that reads a
    device value, statement kinds of lifted format-1 plans: re-plan those first).  `times`: {first result name: device ms}."""
    sts = copy.deepcopy(plan["statements"])
    times = times or {}
    K = max(1, min(int(lanes), 4))
    outputs = set(plan["outputs"])
    n = len(sts)
    for st in sts:
        if st["op"] not in ("call", "reserve", "chview") + _HOST_OPS:
            return None
    # ---- 1. values, aliases, producers
    root = {}                 # value name -> the name whose buffer it lives in
    producer = {}             # value name -> statement index
    writers = {}              # root name -> [statement indices that write into its buffer]
    device = [False] * n      # issues a kernel
    reads = [None] * n
    host_vals = set()
    for i, st in enumerate(sts):
        r = _refs(st.get("args", st.get("in")), [])
        if st["op"] == "chview":
            r = [st["src"]]
        if "window" in st:
            r = r + [st["window"]["of"]] + list(st.get("after", []))
        reads[i] = r
        if st["op"] in _HOST_OPS:
            if any(x in producer and x not in host_vals for x in r):
                return None   # shape arithmetic on a device value: a host read in the middle of the plan
            for o in st["out"]:
                host_vals.add(o)
                producer[o] = i
            continue
        for o in st["out"]:
            producer[o] = i
        if st["op"] == "chview":
            for o in st["out"]:
                root[o] = root.get(st["src"], st["src"])
        elif "window" in st:
            enc = st["window"]["of"]
            for o in st["out"]:
                root[o] = root.get(enc, enc)
            writers.setdefault(root.get(enc, enc), []).append(i)
            device[i] = True
        elif st["op"] == "call" and (st.get("bufs", 1) == 0 or st.get("fn") in _VIEW_FNS):
            src = next((x for x in r if x not in host_vals), None)
            for o in st["out"]:
                root[o] = root.get(src, src) if src is not None else o
        elif st["op"] == "reserve":
            for o in st["out"]:
                root[o] = o
                writers.setdefault(o, [])
        else:
            device[i] = True
            for o in st["out"]:
                root[o] = o
                writers.setdefault(o, []).append(i)
    for name in list(plan["inputs"]):
        root.setdefault(name, name)
    # ---- dependencies between DEVICE statements (views are transparent)
    deps = [set() for _ in range(n)]
    touched = [set() for _ in range(n)]   # roots a device statement reads (for the buffer accesses)
    for i, st in enumerate(sts):
        if not device[i]:
            continue
        for x in reads[i]:
            if x in host_vals:
                continue
            rt = root.get(x, x)
            touched[i].add(rt)
            for w in writers.get(rt, []):
                if w < i:
                    deps[i].add(w)
        if st.get("may_alias"):            # a transpose that may turn out to be a view of its operand at run time
            src = next((x for x in reads[i] if x not in host_vals), None)
            if src is not None:
                st["_alias_of"] = root.get(src, src)
    # readers through a may_alias result also read the source's buffer
    alias_of = {o: st["_alias_of"] for st in sts if "_alias_of" in st for o in st["out"]}

    def alias_chain(rt):     # a transpose of a transpose that may both be views: every buffer the value may live in
        seen = []
        while rt in alias_of and alias_of[rt] not in seen and alias_of[rt] != rt:
            rt = alias_of[rt]
            seen.append(rt)
        return seen
    for i in range(n):
        if device[i]:
            for rt in list(touched[i]):
                touched[i].update(alias_chain(rt))
    # ---- 2. list scheduling
    def cost(i):
        o = sts[i]["out"][0] if sts[i].get("out") else None
        return float(times.get(o, default_ms))
    free = [0.0] * K
    finish = [0.0] * n
    lane_of = [-1] * n
    for i in range(n):
        if not device[i]:
            continue
        ready = max([finish[d] for d in deps[i]] + [0.0])
        pref = 0
        if deps[i]:
            pref = lane_of[max(deps[i], key=lambda d: finish[d])]
        best, best_start = pref, max(ready, free[pref])
        for l in range(K):
            if l == pref:
                continue
            start = max(ready + 0.002, free[l])        # an event edge is not free
            if start + min_gain_ms < best_start:
                best, best_start = l, start
        lane_of[i] = best
        finish[i] = best_start + cost(i)
        free[best] = finish[i]
    sequential = sum(cost(i) for i in range(n) if device[i])
    # ---- 3. events + vector clocks
    seq = [0] * n                  # position of the statement on its lane (1-based)
    count = [0] * K
    vc_lane = [[0] * K for _ in range(K)]          # vc_lane[l][m]: lane l's tail has seen lane m up to this sequence number
    vc_at = [None] * n
    consumers_elsewhere = [False] * n
    for i in range(n):
        if device[i]:
            for d in deps[i]:
                if lane_of[d] != lane_of[i]:
                    consumers_elsewhere[d] = True
    event_of = {}
    waits = [[] for _ in range(n)]
    vc_after = [None] * n
    for i in range(n):
        if not device[i]:
            continue
        l = lane_of[i]
        for d in sorted(deps[i], key=lambda d: -seq[d]):
            m = lane_of[d]
            if m != l and vc_lane[l][m] < seq[d]:
                waits[i].append(event_of[d])
                for k in range(K):                      # the waiting lane learns everything the producer knew when it finished
                    vc_lane[l][k] = max(vc_lane[l][k], vc_after[d][k])
        count[l] += 1
        seq[i] = count[l]
        vc_at[i] = list(vc_lane[l])                     # what has happened-before statement i (its own lane: everything issued earlier)
        vc_at[i][l] = seq[i] - 1
        vc_lane[l][l] = seq[i]
        vc_after[i] = list(vc_lane[l])
        if consumers_elsewhere[i]:
            event_of[i] = len(event_of)
    # ---- 4. slots under happens-before
    accesses = {}      # root name -> [(lane, seq)] of every device statement that reads or writes its buffer
    last_touch = {}    # root name -> last statement index (plan order) that touches it
    for i in range(n):
        if not device[i]:
            continue
        roots = set(touched[i])
        for o in sts[i]["out"]:
            roots.add(root.get(o, o))
        if "window" in sts[i]:
            roots.add(root.get(sts[i]["window"]["of"], sts[i]["window"]["of"]))
        for rt in roots:
            accesses.setdefault(rt, []).append((lane_of[i], seq[i]))
            last_touch[rt] = i
    pinned = {root.get(o, o) for o in outputs} | {root.get(x, x) for x in plan["inputs"]}
    for rt in list(pinned):    # an output that may be a view of its operand at run time keeps the operand's buffer alive too
        pinned.update(alias_chain(rt))
    slot_value = {}    # slot id -> root name it holds
    n_slots = 0
    recent = []        # roots read by the last few device statements (the sequential allocator's guard, kept)

    def happened_before(rt, i):
        return all(vc_at[i][m] >= s for m, s in accesses.get(rt, []))
    for i, st in enumerate(sts):
        needs = st["op"] == "reserve" or (st["op"] == "call" and device[i] and "window" not in st and st.get("bufs", 1) > 0)
        if st["op"] == "reserve":
            # a reserved buffer is written by its windows' statements: it must be free of its previous value for ALL of them; take a slot
            # whose previous value happened-before the FIRST device statement after the reserve on every lane -- conservatively: a fresh
            # slot unless the previous value happened-before every writer
            ws = writers.get(st["out"][0], [])
            cands = [s for s, v in sorted(slot_value.items()) if v not in pinned and last_touch.get(v, -1) < i and ws
                     and all(happened_before(v, w) for w in ws)]
            pick = cands[0] if cands else n_slots
            n_slots = max(n_slots, pick + 1)
            slot_value[pick] = st["out"][0]
            st["slots"] = ["buf_%d" % pick]
            continue
        if not needs:
            continue
        busy = {rt for rec in recent[-5:] for rt in rec} | touched[i]
        st["slots"] = []
        names = st["out"]
        for b in range(st["bufs"]):
            pick = None
            for s, v in sorted(slot_value.items()):
                if v in pinned or v in busy or last_touch.get(v, -1) >= i:
                    continue
                if happened_before(v, i):
                    pick = s
                    break
            if pick is None:
                pick, n_slots = n_slots, n_slots + 1
            st["slots"].append("buf_%d" % pick)
            if b < len(names):
                slot_value[pick] = names[b]
            else:   # a work buffer beyond the results: in use by this statement only
                slot_value[pick] = "%s__work%d" % (names[0], b)
                accesses[slot_value[pick]] = [(lane_of[i], seq[i])]
                last_touch[slot_value[pick]] = i
            busy.add(slot_value[pick])
        recent.append(set(touched[i]))
    # ---- 5. emit
    used = sorted({lane_of[i] for i in range(n) if device[i]})
    for i, st in enumerate(sts):
        st.pop("_alias_of", None)
        if device[i]:
            st["lane"] = lane_of[i]
            if waits[i]:
                st["wait"] = waits[i]
            if i in event_of:
                st["record"] = event_of[i]
    # the last statement of every side lane: lane 0 waits for it before anything reads the outputs
    tails = []
    for l in used:
        if l == 0:
            continue
        t = max(i for i in range(n) if device[i] and lane_of[i] == l)
        if t not in event_of:
            event_of[t] = len(event_of)
            sts[t]["record"] = event_of[t]
        tails.append(event_of[t])
    sts.append({"op": "join", "out": [], "wait": tails})
    # ... and the FIRST statement of every side lane waits for a point on lane 0 taken before anything of this run was issued.  Without
    # it a statement that reads only plan inputs or weights (no producer inside the run: no wait of its own) could, run eagerly,
    # (1) read a device input the caller produced on lane 0's stream before it is ready and (2) on a back-to-back second run
    # overwrite a re-used slot that lane 0's tail of the PREVIOUS run still reads -- the final join orders lane 0 after the side
    # lanes, not the reverse.  In a recorded graph the event stands for no node: nothing is added there.
    if len(used) > 1 or (used and used[0] != 0):
        start = len(event_of)
        event_of["start"] = start
        sts.insert(0, {"op": "join", "out": [], "wait": [], "record": start})
        for l in used:
            if l != 0:
                f = min(i for i in range(n) if device[i] and lane_of[i] == l)
                sts[f + 1]["wait"] = [start] + list(sts[f + 1].get("wait", []))
    new = dict(plan)
    new["statements"] = sts
    new["slots"] = ["buf_%d" % s for s in range(n_slots)]
    makespan = max([finish[i] for i in range(n) if device[i]] + [0.0])
    new["dag"] = {"lanes": len(used), "events": len(event_of), "statements_by_lane": [sum(1 for i in range(n) if device[i] and lane_of[i] == l) for l in range(K)],
                  "modelled_sequential_ms": round(sequential, 4), "modelled_makespan_ms": round(makespan, 4),
                  "slots_before": len(plan.get("slots", [])), "slots_after": n_slots}
    return new
