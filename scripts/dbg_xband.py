import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if __name__ == "__main__":
    from tests import test_multirank_gpu as T
    nd = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    grid = (1440, 1080, 30, 27) if len(sys.argv) < 3 else tuple(int(v) for v in sys.argv[2].split(","))
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    try:
        T._run(world, "tripole", *grid, ndte=nd, env=dict({"TEST_DEBUG_LOC": "1"}, **({"EVPK_ZONE_M": os.environ["ZM"]} if "ZM" in os.environ else {})), xp="ipc")
        print("OK")
    except AssertionError as e:
        print("ASSERT", str(e)[:300])
