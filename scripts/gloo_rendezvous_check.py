"""Two local processes rendezvous over torch.distributed "gloo" at 127.0.0.1 exactly as bench.py's ranks do (the GPU box's
hostname may not resolve): prints what every rank received."""
import os, socket, sys
import torch.multiprocessing as mp


def work(rank, world, port, ifname):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if ifname:
        os.environ["GLOO_SOCKET_IFNAME"] = ifname
    import torch.distributed as dist
    dist.init_process_group("gloo")
    box = [b"x" * 128 if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    from torch.distributed.distributed_c10d import _get_default_store
    st = _get_default_store()
    st.set(f"k{rank}", b"v%d" % rank)
    st.wait([f"k{1 - rank}"])
    print(f"rank {rank}: id {len(box[0])} bytes, store peer value {st.get(f'k{1 - rank}')}, ifname {ifname!r}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    for ifname in ("lo", None):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        try:
            mp.spawn(work, args=(2, port, ifname), nprocs=2, join=True)
        except Exception as ex:
            print(f"ifname {ifname!r}: FAILED {type(ex).__name__}: {str(ex)[:300]}")
