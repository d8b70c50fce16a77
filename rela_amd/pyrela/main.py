"""Ape-X training entry point on the MI355X-native `rela` module (counterpart of pyrela/main.py).

Flag names, defaults and the loop shape follow pyrela/main.py:23-82,197-251: actors fill the
replay from C++ threads, the learner samples / steps / updates priorities, actor weights are
re-published every --actor_sync_freq updates, the target net every --num_update_between_sync.
Differences, all on purpose:
  * envs are synthetic (create_env.py); --game only names the run;
  * the priority stays on the GPU between loss() and update_priority() (no per-step host sync);
  * several --act_device values: by default one PROCESS per device (train_multi, rela_amd/parallel.py);
    --single_process 1 keeps the reference's own wiring (main.py:131-136,155,166): one ModelLocker per act device in
    THIS process, threads dealt round-robin onto them, all feeding the one replay object, which then owns one
    partition per locker (rela module, ReplayParts).  The eval locker lives on "cpu" as in main.py:116 and
    --num_eval_game > 0 runs the reference's eval cadence (main.py:264-278) after every epoch.
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
    p.add_argument("--num_eval_game", type=int, default=0, help="eval threads after every epoch (pyrela/main.py:264-278); 0 = off")
    p.add_argument("--exchange", type=str, default="native",
                   help="several processes (train_multi): native = replay partitions and flat weight buffers mapped through "
                        "HIP IPC, the learner gathers the sampled rows itself over xGMI (rela_amd/parallel.py); packed = one "
                        "packed gather / two broadcasts through the collective")
    p.add_argument("--single_process", type=int, default=0,
                   help="several --act_device values in ONE process, as the reference wires them (main.py:131-136)")
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

    # create eval locker here, as upstream (main.py:115-116): replicas on the host, device string "cpu"
    eval_locker = rela.ModelLocker([type(agent).clone(agent, "cpu")], "cpu")
    act_devices = args.act_device.split(",")
    if len(act_devices) != 1 and not getattr(args, "single_process", 0):  # (see __main__: train_multi)
        raise SystemExit("several --act_device values: use --single_process 1 (the reference's wiring) or "
                         "rela_amd.pyrela.main.train_multi (one process per GPU)")
    lockers = [rela.ModelLocker([agent, agent, agent], d) for d in act_devices]  # 3 weight versions per device (:131-136)

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
        watch = _LearnerWatch(learner, args.train_device)
        for batch_idx in range(args.epoch_len):
            num_update = batch_idx + epoch * args.epoch_len
            if num_update % args.num_update_between_sync == 0:
                if learner is not None:
                    learner.sync_target_with_online()
                else:
                    agent.sync_target_with_online()
            if num_update % args.actor_sync_freq == 0:
                watch.poll()
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
            watch.add(loss.detach())
        torch.cuda.synchronize()
        mean_loss = watch.mean_loss()
        dt = time.time() - t0
        print("epoch: %d, time: %.1fs, loss: %.5f%s" % (epoch, dt, mean_loss, (
            " (%d steps skipped after a learner timeout)" % watch.skipped) if watch.skipped else ""))
        rates = tach.lap(actors, replay_buffer, args.epoch_len * args.batchsize)
        history.append(dict(epoch=epoch, seconds=dt, train=rates[0], act=rates[1], buffer_add=rates[2],
                            loss=mean_loss, skipped_steps=watch.skipped))
        if getattr(args, "num_eval_game", 0) > 0:  # main.py:264-278
            from rela_amd.pyrela.eval import evaluate

            context.pause()
            if learner is not None:
                agent.online_net.load_state_dict(learner.state_dict("online"))
                agent.target_net.load_state_dict(learner.state_dict("target"))
            eval_locker.update_model(agent)
            actor_cls = rela.R2D2Actor if args.algo == "r2d2" else rela.DQNActor
            score = evaluate(args.num_eval_game, eval_locker, actor_cls, epoch * args.num_eval_game + 1, args.episode_len, 0)
            print("epoch %d, eval score: %f" % (epoch, score))
            history[-1]["eval_score"] = score
            context.resume()
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


# ---- the reference's multi-GPU layout (pyrela/main.py:131-166), one process per GPU -------------------------
def _exchange_backend(args):
    """RCCL when every rank has its own GPU; gloo with host-side exchange buffers when ranks share a card
    (the one-GPU rehearsal: RCCL refuses two ranks on one device)."""
    devs = [args.train_device] + args.act_device.split(",")
    shared = len(set(devs)) < len(devs)
    return ("gloo", "cpu") if shared else ("nccl", None)


class _RelaFFPartition:
    """the `rela` module's FFPrioritizedReplay as a partition of rela_amd.parallel"""

    def __init__(self, replay, device):
        self.replay, self.device = replay, device

    def sample(self, n):
        b, _ = self.replay.sample(n, self.device)
        raw_w, part_sum, size = self.replay.last_sample_raw()
        fields = {"s": b.obs["s"], "next_s": b.next_obs["s"], "eps": b.obs["eps"], "next_eps": b.next_obs["eps"],
                  "legal_move": b.obs["legal_move"], "next_legal_move": b.next_obs["legal_move"], "a": b.action["a"],
                  "reward": b.reward, "terminal": b.terminal.to(torch.uint8), "bootstrap": b.bootstrap}
        return fields, raw_w, part_sum, size

    def update_priority(self, p):
        self.replay.update_priority(p.to(self.device))


class _RelaRNNPartition:
    """the `rela` module's RNNPrioritizedReplay (one slot = one sequence, time-major batches) as a partition"""

    def __init__(self, replay, device):
        self.replay, self.device = replay, device

    def sample(self, n):
        b, _ = self.replay.sample(n, self.device)
        raw_w, part_sum, size = self.replay.last_sample_raw()
        fields = {"s": b.obs["s"], "eps": b.obs["eps"], "legal_move": b.obs["legal_move"], "a": b.action["a"],
                  "reward": b.reward, "terminal": b.terminal.to(torch.uint8), "bootstrap": b.bootstrap,
                  "h0": b.h0["h0"], "c0": b.h0["c0"], "seq_len": b.seq_len}
        return fields, raw_w, part_sum, size

    def update_priority(self, p):
        self.replay.update_priority(p.to(self.device))


def _multi_worker(rank, world, args, port, results):
    import torch.distributed as dist

    from rela_amd.learner import HipApexLearner, HipR2D2Learner, ffnet_flat_layout, lstmnet_flat_layout
    from rela_amd.parallel import (NativePartitionedReplay, NativePartitionServer, PartitionedReplay, PartitionServer,
                                   _ModuleNativePartition, ff_batch_namespace, ff_field_specs, rnn_batch_namespace,
                                   rnn_field_specs)

    native = getattr(args, "exchange", "native") == "native"

    act_devices = args.act_device.split(",")
    G = len(act_devices)
    backend, exch = _exchange_backend(args)
    my_device = args.train_device if rank == 0 else act_devices[rank - 1]
    torch.cuda.set_device(torch.device(my_device))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(my_device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    exch_device = exch or my_device
    torch.manual_seed(args.seed + 2)  # identical initial weights on every rank
    num_action = create_env.get_num_action(args.game)
    r2d2 = args.algo == "r2d2"
    if r2d2:  # BASELINE C4: sequence replay partitions, LSTM nets (pyrela/main.py:98-109)
        agent = R2D2Agent(lambda dev: AtariLSTMNet(dev, num_action), my_device, args.multi_step, args.gamma, args.eta,
                          args.seq_len, args.seq_burn_in, args.same_hid).to(my_device)
        specs = rnn_field_specs(num_action, args.seq_burn_in + args.seq_len + args.multi_step)
        layout, total = lstmnet_flat_layout(num_action)
        to_namespace, learner_cls = rnn_batch_namespace, HipR2D2Learner
    else:
        agent = ApexAgent(lambda: AtariFFNet(num_action), args.multi_step, args.gamma).to(my_device)
        specs = ff_field_specs(num_action)
        layout, total = ffnet_flat_layout(num_action)
        to_namespace, learner_cls = ff_batch_namespace, HipApexLearner
    assert args.batchsize % G == 0 and args.num_thread % G == 0
    if rank == 0:
        learner = learner_cls.from_agent(agent, args.batchsize, lr=args.lr, eps=args.eps, grad_clip=args.grad_clip)
        # scheduled exchange: command words (a host synchronisation on every rank) only with the weight publish every
        # actor_sync_freq steps; the sample / update_priority pairs in between follow the announced cycle
        if native:  # partitions and flat buffers mapped through HIP IPC: rows and weights never enter a collective
            replay = NativePartitionedReplay(specs, args.batchsize, args.importance_exponent, exch_device, scheduled=True,
                                             flats=(learner.flat()[0], learner.flat_target()), data_device=my_device)
        else:
            replay = PartitionedReplay(specs, args.batchsize, args.importance_exponent, exch_device, scheduled=True)
        total_updates = args.num_epoch * args.epoch_len
        history = []
        for epoch in range(args.num_epoch):
            t0 = time.time()
            watch = _LearnerWatch(learner, my_device)
            for batch_idx in range(args.epoch_len):
                num_update = batch_idx + epoch * args.epoch_len
                if num_update % args.num_update_between_sync == 0:
                    learner.sync_target_with_online()
                if num_update % args.actor_sync_freq == 0:  # ONE broadcast per flat buffer instead of load_state_dict
                    watch.poll()
                    if native:  # (the actors read the mapped buffers themselves)
                        replay.publish(learner.flat()[0], learner.flat_target(),
                                       steps=min(args.actor_sync_freq, total_updates - num_update))
                    else:
                        replay.publish(learner.flat()[0].to(exch_device), learner.flat_target().to(exch_device),
                                       steps=min(args.actor_sync_freq, total_updates - num_update))
                fields, weight = replay.sample()
                batch = to_namespace({k: v.to(my_device) for k, v in fields.items()})
                loss, priority = learner.step(batch, weight.to(my_device))
                replay.update_priority(priority)
                watch.add(loss[0])
            torch.cuda.synchronize()
            mean_loss = watch.mean_loss()
            dt = time.time() - t0
            history.append(dict(epoch=epoch, seconds=dt, train=args.epoch_len * args.batchsize / dt,
                                loss=mean_loss, skipped_steps=watch.skipped))
            print("epoch: %d, time: %.1fs, loss: %.5f, train: %.1f samples/s" % (
                epoch, dt, history[-1]["loss"], history[-1]["train"]), flush=True)
        replay.stop()
        if native:
            replay.close()
        counts = torch.zeros(2, dtype=torch.float64, device=exch_device)
        dist.all_reduce(counts)
        results.put(dict(history=history, act=float(counts[0]), buffer_add=float(counts[1])))
    else:
        g = rank - 1
        locker = rela.ModelLocker([agent, agent, agent], my_device)
        if native:  # the learner maps this partition: fields as 8 GB chunks, which travel at any size (include/rela_amd.h)
            rela.set_replay_chunk_bytes(8 << 30)
        part = (rela.RNNPrioritizedReplay if r2d2 else rela.FFPrioritizedReplay)(
            args.replay_buffer_size // G, args.seed + g, args.priority_exponent, args.importance_exponent, args.prefetch)
        eps_all = utils.generate_eps(args.act_base_eps, args.act_eps_alpha, args.num_thread * args.num_game_per_thread)
        # the reference deals thread t to device t % G (main.py:155,166): this rank runs those threads
        threads = [t for t in range(args.num_thread) if t % G == g]
        K = args.num_game_per_thread
        eps = [e for t in threads for e in eps_all[t * K:(t + 1) * K]]
        if r2d2:
            make_actor = lambda i: rela.R2D2Actor(locker, args.multi_step, K, args.gamma, args.seq_len, args.seq_burn_in,
                                                  part)
        else:
            make_actor = lambda i: rela.DQNActor(locker, args.multi_step, K, args.gamma, part)
        context, games, actors = create_env.create_train_env(args.seed + 7919 * g, eps, args.episode_len, len(threads), K,
                                                             make_actor)
        context.start()
        while part.size() < max(args.burn_in_frames // G, args.batchsize // G):
            time.sleep(0.05)

        def on_weights(on_flat, tg_flat):
            sd = {}
            for prefix, flat in (("online_net.", on_flat), ("target_net.", tg_flat)):
                for key, shape, off in layout:
                    n = 1
                    for d in shape:
                        n *= d
                    sd[prefix + key] = flat[off:off + n].view(shape)
            agent.load_state_dict(sd)
            locker.update_model(agent)

        if native:
            srv = NativePartitionServer(_ModuleNativePartition(part, my_device), specs, args.batchsize,
                                        args.importance_exponent, exch_device, on_weights=on_weights, scheduled=True,
                                        data_device=my_device)
        else:
            srv = PartitionServer((_RelaRNNPartition if r2d2 else _RelaFFPartition)(part, my_device), specs, args.batchsize,
                                  args.importance_exponent,
                                  exch_device, flat_sizes=(total, total), on_weights=on_weights, scheduled=True)
        srv.serve_forever()
        if native:
            srv.close()
        counts = torch.tensor([float(utils.total_acts(actors)), float(part.num_add())], dtype=torch.float64,
                              device=exch_device)
        dist.all_reduce(counts)
        context.terminate()
        context.resume()
        t0 = time.time()
        while not context.terminated() and time.time() - t0 < 60:
            if part.size() >= args.batchsize // G:
                _, w = part.sample(args.batchsize // G, my_device)
                part.update_priority(w)
            time.sleep(0.01)
    dist.barrier()
    dist.destroy_process_group()



class _LearnerWatch:
    """Keeps an epoch's loss statistics honest when a HIP learner's persistent kernels give up on a grid barrier
    (rela_r2d2_learner_check): after a timeout the word stays set, every later persistent launch leaves at its first
    barrier and every apply() is skipped on the device until the word is read.  The word is therefore polled at a BOUNDED
    interval -- with every weight publish (actor_sync_freq steps, where the loop synchronises anyway) and at the end
    of an epoch -- instead of once per epoch.  A hit clears the word and switches the learner to its per-step launches
    (inside check()), the steps since the last clean poll are counted as skipped and their losses are left out of the
    epoch's average; their priorities (<= actor_sync_freq batches, from a forward that did not finish) were already
    written to the replay and are overwritten the next time those slots are sampled."""

    def __init__(self, learner, device):
        self.learner = learner
        self.good = torch.zeros((), device=device)
        self.pending = torch.zeros((), device=device)
        self.n_good = self.n_pending = self.skipped = 0

    def add(self, loss):
        self.pending += loss
        self.n_pending += 1

    def poll(self):
        ok = True
        if self.learner is not None and hasattr(self.learner, "check") and self.n_pending:
            try:
                self.learner.check()
            except RuntimeError as e:
                ok = False
                self.skipped += self.n_pending
                print("WARNING: %s -- %d learner steps since the last clean poll are excluded from the loss average "
                      "(%d skipped so far)" % (e, self.n_pending, self.skipped), flush=True)
        if ok:
            self.good += self.pending
            self.n_good += self.n_pending
        self.pending.zero_()
        self.n_pending = 0
        return ok

    def mean_loss(self):
        self.poll()
        return float(self.good) / max(self.n_good, 1)


def train_multi(args):
    """--act_device cuda:1,cuda:2,...: one actor process per act device, each with its replay partition, plus
    the learner process on --train_device (rela_amd/parallel.py).  Processes are spawned BEFORE any GPU call."""
    import socket

    import torch.multiprocessing as mp

    if args.algo not in ("apex", "r2d2"):
        raise SystemExit("--algo must be apex or r2d2")
    G = len(args.act_device.split(","))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    results = ctx.Queue()
    procs = [ctx.Process(target=_multi_worker, args=(r, G + 1, args, port, results)) for r in range(G + 1)]
    for p in procs:
        p.start()
    # No global deadline (a default run lasts many hours): poll for the learner's result and watch every rank.  The
    # first rank that dies takes the job down -- its peers would otherwise hang in their collectives.
    import queue

    def stop_all():
        for q in procs:
            if q.is_alive():
                q.terminate()
        for q in procs:
            q.join(timeout=10)

    res = None
    while res is None:
        try:
            res = results.get(timeout=5)
        except queue.Empty:
            dead = [(r, q.exitcode) for r, q in enumerate(procs) if q.exitcode not in (None, 0)]
            if dead:
                stop_all()
                raise SystemExit("rank %d exited with code %s before the run finished; the other ranks were stopped" % dead[0])
            if all(q.exitcode == 0 for q in procs):
                try:  # the learner may have put its result and every rank exited between the timed-out get and here
                    res = results.get(timeout=1)
                except queue.Empty:
                    raise SystemExit("every rank exited without a result")
    for r, q in enumerate(procs):
        q.join(timeout=120)
        if q.exitcode is None:  # still running two minutes after the result: stop it, the run itself succeeded
            q.terminate()
            q.join(timeout=10)
        elif q.exitcode != 0:
            stop_all()
            raise SystemExit("rank %d exited with code %s" % (r, q.exitcode))
    return res


if __name__ == "__main__":
    _args = parse_args()
    if (len(_args.act_device.split(",")) > 1 or _args.act_device != _args.train_device) and not _args.single_process:
        print(train_multi(_args))
    else:
        train(_args)
