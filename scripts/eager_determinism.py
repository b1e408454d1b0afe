"""Do all eager steps of the benchmark loop give the same rows?  Every slot holds the same images, so every step must reproduce
the same [B,16] block bit for bit.  Runs the bench's in-flight loop for N steps and lists the steps that deviate from the majority.
usage: eager_determinism.py [steps] [inflight] [two_streams 0/1] [stash: none|feats|pose|head|all]
`stash` keeps the named intermediate tensors of a step alive until the slot's next step (no extra GPU work): if a deviation
disappears, a tensor of that group is freed on the host while a kernel on ANOTHER stream still reads or writes it."""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nopesac_amd import ops, runner  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 4
two = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
stash_what = sys.argv[4] if len(sys.argv) > 4 else "none"
B, K, nq = 32, 32, 50
dev = torch.device("cuda:0")
model = bench.build_model(dev, nq, "bfloat16")
model.two_streams = two
routing = os.path.join(ROOT, "profiles", "routing_r5.json")
if os.path.exists(routing):
    ops.TUNER.load(routing)
model.autotune(B)
raw = torch.randint(0, 256, (2 * B, 3, 480, 640), generator=torch.Generator().manual_seed(1000)).float().to(dev)
raws = [raw] + [raw.clone() for _ in range(inflight - 1)]
forced = bench.make_forced(B, K, nq, dev, 7)
loop = runner.InflightLoop(inflight, B, dev, 1)
stash = {}
cur_slot = [0]
head = model.camera_head_list[0]
if stash_what in ("feats", "all"):
    orig_bb = model.backbone.forward

    def bb(*a, **k):
        out = orig_bb(*a, **k)
        stash[("feats", cur_slot[0])] = out
        return out
    model.backbone.forward = bb
if stash_what in ("pose", "all"):
    orig_pose = head.initial_pose

    def ip(*a, **k):
        out = orig_pose(*a, **k)
        stash[("pose", cur_slot[0])] = out
        return out
    head.initial_pose = ip
if stash_what in ("head", "all"):
    orig_head = model.sem_seg_head.forward

    def hd(*a, **k):
        out = orig_head(*a, **k)
        stash[("head", cur_slot[0])] = out
        return out
    model.sem_seg_head.forward = hd


dup = {}
orig_sm = ops.ransac_score_maps


def sm_twice(*a, **k):
    """The score-map kernel twice on the same inputs: do the two launches of ONE step agree?"""
    o1 = orig_sm(*a, **k)
    o2 = orig_sm(*a, **k)
    dup[cur_slot[0]] = (o1["normal_score"] - o2["normal_score"]).abs().amax(dim=(1, 2)) + (o1["param_score"] - o2["param_score"]).abs().amax(dim=(1, 2))
    return o1


if os.environ.get("DUP_SCORE_MAPS"):
    ops.ransac_score_maps = sm_twice


def device_step(slot):
    cur_slot[0] = slot
    d = model.forward_tensors(None, B, 480, 640, forced=forced, raw_images=raws[slot])
    cam = d["cam"]
    rows = runner.metric_rows(cam["cameras"]["camera"][0], cam["cameras"]["camera"][1], cam["n1"], cam["n2"], cam["m"], 0, nonfinite=cam.get("nonfinite"))
    # per-pair sums of the refine stage's intermediates ride in the spare columns: where does a deviation start?
    r = cam["refine"]
    mp = r["maps"]
    rows[:, 7] = cam["cameras"]["camera_initRec"][0].sum(1) + cam["cameras"]["camera_initRec"][1].sum(1) + d["sel"]["feats"][:B].sum((1, 2))   # + plane-head query embeddings
    rows[:, 8] = r["geo_local"].sum((1, 2)) + r["sig"].sum(1)
    rows[:, 9] = mp["rots_all"].sum((1, 2))            # normalised hypothesis rotations: decoder_rot / rot2 / rots chains
    rows[:, 10] = mp["trans_all"].sum((1, 2))          # hypothesis translations: geo_proj_s2 / decoder_tran / tran2 / trans chains
    rows[:, 11] = mp["normal_score"].sum((1, 2))
    rows[:, 12] = mp["param_score"].sum((1, 2))
    rows[:, 13] = r["score_rot"].sum(1)
    rows[:, 14] = r["score_trans"].sum(1)
    rows[:, 15] = cam["init_feats"][0].sum(1) + cam["init_feats"][1].sum(1)      # AIM features (x_bcast of the second chains)
    if slot in dup:
        rows[:, 15] = dup[slot]                            # |first launch - second launch| of the score maps (0 = they agree)
    return None, rows


hist, results = [], []
for i in range(steps):
    slot = i % inflight
    if loop.done[slot] is not None and len(hist) >= inflight:
        j, _ = hist[-inflight]
        loop.done[slot].synchronize()
        results.append((j, loop.host_bufs[slot].clone()))
    loop.step(i, device_step)
    hist.append((i, slot))
loop.barrier()
for j, slot in hist[-inflight:]:
    results.append((j, loop.host_bufs[slot].clone()))
results.sort(key=lambda t: t[0])
blocks = [b for _, b in results]
keys = [b.numpy().tobytes() for b in blocks]
major, cnt = Counter(keys).most_common(1)[0]
ref = blocks[keys.index(major)]
bad = []
for i, (b, k) in enumerate(zip(blocks, keys)):
    if k != major:
        names = ["camera", "initRec+query_feats", "geo", "rots_all", "trans_all", "normal_score", "param_score", "score_rot", "score_trans", "aim_feats_or_dup_diff"]
        cols = [slice(0, 7)] + [slice(c, c + 1) for c in range(7, 16)]
        dev_ = ["%s %.2g" % (n, float((b[:, c] - ref[:, c]).abs().max())) for n, c in zip(names, cols) if float((b[:, c] - ref[:, c]).abs().max()) > 0]
        bad.append("step %d slot %d: %s; pairs %d" % (i, i % inflight, ", ".join(dev_), int(((b - ref).abs().amax(dim=1) > 0).sum())))
print("inflight %d two_streams %s stash %s: %d steps, %d identical to the majority" % (inflight, two, stash_what, len(blocks), cnt))
for l in bad[:12]:
    print("   " + l)
