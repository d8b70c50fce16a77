"""Ape-X training entry point on the MI355X-native `rela` module (counterpart of pyrela/main.py).

Flag names, defaults and the loop shape follow pyrela/main.py:23-82,197-251: actors fill the
replay from C++ threads, the learner samples / steps / updates priorities, actor weights are
re-published every --actor_sync_freq updates, the target net every --num_update_between_sync.
Differences, all on purpose:
  * envs are synthetic (create_env.py); --game only names the run; evaluation is out of scope;
  * the priority stays on the GPU between loss() and update_priority() (no per-step host sync);
  * one replay partition per actor GPU: several --act_device values need one process each.
"""
import argparse
import os
import pprint
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from rela_amd.pyrela import create_env, utils  # noqa: E402
from rela_amd.pyrela.apex import ApexAgent  # noqa: E402
from rela_amd.pyrela.net import AtariFFNet, AtariLSTMNet  # noqa: E402
from rela_amd.pyrela.r2d2 import R2D2Agent  # noqa: E402

rela = create_env.rela


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Ape-X on synthetic Atari-shaped envs (MI355X)")
    p.add_argument("--save_dir", type=str, default="exps/exp1")
    p.add_argument("--multi_step", type=int, default=3)
    p.add_argument("--algo", type=str, default="apex", help="apex/r2d2")
    p.add_argument("--seq_burn_in", type=int, default=40)
    p.add_argument("--seq_len", type=int, default=80)
    p.add_argument("--eta", type=float, default=0.9)
    p.add_argument("--same_hid", type=int, default=0)
    p.add_argument("--game", type=str, default="synthetic")
    p.add_argument("--seed", type=int, default=10002)
    p.add_argument("--max_frame", type=int, default=108000)
    p.add_argument("--episode_len", type=int, default=200, help="synthetic episode length (SURVEY 8d)")
    p.add_argument("--gamma", type=float, default=0.997)
    p.add_argument("--lr", type=float, default=6.25e-5)
    p.add_argument("--eps", type=float, default=1.5e-4)
    p.add_argument("--grad_clip", type=float, default=40)
    p.add_argument("--batchsize", type=int, default=512)
    p.add_argument("--num_epoch", type=int, default=3000)
    p.add_argument("--epoch_len", type=int, default=1000)
    p.add_argument("--num_update_between_sync", type=int, default=2500)
    p.add_argument("--train_device", type=str, default="cuda:0")
    p.add_argument("--burn_in_frames", type=int, default=80000)
    p.add_argument("--replay_buffer_size", type=int, default=int(2e6))
    p.add_argument("--prefetch", type=int, default=1)
    p.add_argument("--priority_exponent", type=float, default=0.6)
    p.add_argument("--importance_exponent", type=float, default=0.4)
    p.add_argument("--num_thread", type=int, default=40)
    p.add_argument("--num_game_per_thread", type=int, default=20)
    p.add_argument("--act_base_eps", type=float, default=0.4)
    p.add_argument("--act_eps_alpha", type=float, default=7)
    p.add_argument("--act_device", type=str, default="cuda:0")
    p.add_argument("--actor_sync_freq", type=int, default=20)
    p.add_argument("--hip_learner", type=int, default=1,
                   help="run loss / backward / clip / optimiser through the hand-written HIP learner step "
                        "(csrc/learner.hip for apex, csrc/learner_r2d2.hip for r2d2) instead of PyTorch autograd")
    return p.parse_args(argv)


def train(args, on_epoch=None):
    if args.algo not in ("apex", "r2d2"):
        raise SystemExit("--algo must be apex or r2d2")
    torch.manual_seed(args.seed + 2)
    torch.cuda.manual_seed(args.seed + 3)
    pprint.pprint(vars(args))

    num_action = create_env.get_num_action(args.game)
    if args.algo == "r2d2":  # pyrela/main.py:98-109,123-126
        agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, num_action), args.train_device, args.multi_step, args.gamma,
                          args.eta, args.seq_len, args.seq_burn_in, args.same_hid).to(args.train_device)
        optim = torch.optim.Adam(agent.online_net.parameters(), lr=args.lr, eps=args.eps)
        replay_class = rela.RNNPrioritizedReplay
    else:  # :110-122
        agent = ApexAgent(lambda: AtariFFNet(num_action), args.multi_step, args.gamma).to(args.train_device)
        optim = torch.optim.RMSprop(agent.online_net.parameters(), lr=args.lr, eps=args.eps)
        replay_class = rela.FFPrioritizedReplay
    learner = None
    if getattr(args, "hip_learner", 1):  # hand-written HIP learner step (csrc/learner.hip, csrc/learner_r2d2.hip)
        from rela_amd.learner import HipApexLearner, HipR2D2Learner

        cls = HipR2D2Learner if args.algo == "r2d2" else HipApexLearner
        learner = cls.from_agent(agent, args.batchsize, lr=args.lr, eps=args.eps, grad_clip=args.grad_clip)

    act_devices = args.act_device.split(",")
    if len(act_devices) != 1:
        raise SystemExit("one replay partition per actor GPU: launch one process per act device (DESIGN.md §6)")
    lockers = [rela.ModelLocker([agent, agent, agent], d) for d in act_devices]  # 3 weight versions per device

    replay_buffer = replay_class(args.replay_buffer_size, args.seed, args.priority_exponent,
                                 args.importance_exponent, args.prefetch)
    explore_eps = utils.generate_eps(args.act_base_eps, args.act_eps_alpha, args.num_thread * args.num_game_per_thread)
    if args.algo == "r2d2":
        make_actor = lambda i: rela.R2D2Actor(lockers[i % len(lockers)], args.multi_step, args.num_game_per_thread,
                                              args.gamma, args.seq_len, args.seq_burn_in, replay_buffer)
    else:
        make_actor = lambda i: rela.DQNActor(lockers[i % len(lockers)], args.multi_step, args.num_game_per_thread,
                                             args.gamma, replay_buffer)
    print("creating train env")
    context, games, actors = create_env.create_train_env(args.seed, explore_eps, args.episode_len, args.num_thread,
                                                         args.num_game_per_thread, make_actor)
    context.start()
    while replay_buffer.size() < args.burn_in_frames:
        print("warming up replay buffer:", replay_buffer.size())
        time.sleep(1)

    tach = utils.Tachometer()
    history = []
    for epoch in range(args.num_epoch):
        tach.start()
        t0 = time.time()
        loss_sum = torch.zeros((), device=args.train_device)
        for batch_idx in range(args.epoch_len):
            num_update = batch_idx + epoch * args.epoch_len
            if num_update % args.num_update_between_sync == 0:
                if learner is not None:
                    learner.sync_target_with_online()
                else:
                    agent.sync_target_with_online()
            if num_update % args.actor_sync_freq == 0:
                if learner is not None:  # ModelLocker reads the Python model: hand it the current weights
                    agent.online_net.load_state_dict(learner.state_dict("online"))
                    agent.target_net.load_state_dict(learner.state_dict("target"))
                for locker in lockers:
                    locker.update_model(agent)
            batch, weight = replay_buffer.sample(args.batchsize, args.train_device)
            if learner is not None:
                loss, priority = learner.step(batch, weight)
                loss = loss[0].clone()
            else:
                loss, priority = agent.loss(batch, sync_priority=False)
                loss = (loss * weight).mean()
                loss.backward()
                torch.nn.utils.clip_grad_norm_(agent.online_net.parameters(), args.grad_clip)
                optim.step()
                optim.zero_grad()
            replay_buffer.update_priority(priority)
            loss_sum += loss.detach()
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("epoch: %d, time: %.1fs, loss: %.5f" % (epoch, dt, float(loss_sum) / args.epoch_len))
        rates = tach.lap(actors, replay_buffer, args.epoch_len * args.batchsize)
        history.append(dict(epoch=epoch, seconds=dt, train=rates[0], act=rates[1], buffer_add=rates[2],
                            loss=float(loss_sum) / args.epoch_len))
        if on_epoch is not None:
            on_epoch(history[-1])
        print("****************************************")
    context.terminate()
    context.resume()
    while not context.terminated():
        # actor threads may be parked on a full ring (back-pressure): keep draining until they exit
        if replay_buffer.size() >= args.batchsize:
            batch, weight = replay_buffer.sample(args.batchsize, args.train_device)
            replay_buffer.update_priority(weight)
        time.sleep(0.01)
    return history


if __name__ == "__main__":
    train(parse_args())
