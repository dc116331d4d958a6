"""One KD-Lite fine-tune step (pruning/trainer.py:169-194: student forward, teacher forward, KD-Lite
loss, backward, Adam) on a half-pruned DeepLabV3+-R50, section by section with HIP events, with the
library's fused loss and with the same loss written in torch ops beside it.
Usage: python tools/ftbench.py [--steps 10] [--batch 8] [--size 512] [--classes 21]"""
import argparse, copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from spice_amd import engine, nets, synth
from spice_amd.pruning.losses import KDLiteLoss

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--classes", type=int, default=21)
args = ap.parse_args()
dev = torch.device("cuda:0"); C = args.classes
torch.manual_seed(0)
teacher = nets.create_model("DeepLabV3Plus", "resnet50", C).to(dev).eval()
student = copy.deepcopy(teacher)
ignored = [m for m in student.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
engine.MagnitudePruner(student, torch.randn(1, 3, 64, 64, device=dev), pruning_ratio=0.5, ignored_layers=ignored).step()
student.train()
img = torch.from_numpy(synth.normal(7, 1, args.batch * 3 * args.size * args.size).reshape(args.batch, 3, args.size, args.size)).to(dev)
lab = torch.from_numpy(synth.label_map(7, args.batch, args.size, args.size, C)).to(dev)
T, alpha = 4.0, 0.7
fused = KDLiteLoss(T, alpha, num_classes=C).to(dev)

def torch_loss(s, t, y):
    ce = F.cross_entropy(s, y, ignore_index=255)
    valid = (y != 255)
    kl = F.kl_div(F.log_softmax(s / T, 1), F.softmax(t / T, 1), reduction="none").sum(1)
    kd = (kl * valid).sum() / valid.sum().clamp(min=1) * T * T
    return alpha * ce + (1 - alpha) * kd

def run(loss_kind):
    opt = torch.optim.Adam([p for p in student.parameters() if p.requires_grad], lr=1e-5)
    names = ["student_fwd", "teacher_fwd", "loss_fwd", "backward(loss+net)", "adam"]
    acc = {k: 0.0 for k in names}
    wall = 0.0
    for it in range(args.steps + 3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        opt.zero_grad(set_to_none=True)
        ev[0].record()
        if loss_kind == "fused_label_pass_aside":
            fused.prepare_labels(lab)   # the loss's label pass on a stream of its own, beside the forwards
        s = student(img)
        ev[1].record()
        with torch.no_grad():
            t = teacher(img)
        ev[2].record()
        loss = fused(s, t, lab)["total"] if loss_kind.startswith("fused") else torch_loss(s, t, lab)
        ev[3].record()
        loss.backward()
        ev[4].record()
        opt.step()
        ev[5].record()
        torch.cuda.synchronize()
        if it >= 3:
            for k, name in enumerate(names):
                acc[name] += ev[k].elapsed_time(ev[k + 1])
            wall += ev[0].elapsed_time(ev[5])
    out = {k: round(v / args.steps, 3) for k, v in acc.items()}
    out["step_ms"] = round(wall / args.steps, 3)
    return out

res = {"config": {"model": "DeepLabV3Plus_resnet, every prunable group at 50 %", "batch": [args.batch, 3, args.size, args.size],
                  "classes": C, "dtype": "fp32 (the reference trains without autocast)", "steps": args.steps},
       "fused_loss": run("fused"), "fused_loss_label_pass_beside_the_forwards": run("fused_label_pass_aside"),
       "torch_op_loss": run("torch")}
print(json.dumps(res))
