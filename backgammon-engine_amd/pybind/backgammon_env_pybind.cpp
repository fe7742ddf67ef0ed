// backgammon_env_pybind.cpp -- the reference's pybind11 module `backgammon_env` (cppsrc/backgammon_bindings.cpp:41-94) bound to
// the MI355X C ABI (include/bgamd.h) instead of cppsrc/game.cpp: the same classes, method names, argument meaning, return
// types and error behaviour, so reference Python (`import backgammon_env as bg`, model.py:29, train.py:13, benchmark.py:12)
// runs on it unchanged.  No HIP and no torch in this file: every method is one call of the host-argument scalar surface
// `bgamd_game_*` on a one-lane device env.  Built by __graft_entry__.build() with g++ and the installed pybind11:
//   g++ -O2 -std=c++17 -shared -fPIC $(python3 -m pybind11 --includes) -I include backgammon_env_pybind.cpp \
//       -L backgammon-engine_amd -lbgamd -Wl,-rpath,'$ORIGIN/..' -o pybind/backgammon_env$(python3-config --extension-suffix)
// and found by putting backgammon-engine_amd/pybind on sys.path where the reference puts <repo>/build (model.py:26).
// Differences a maintainer should know are those of INTEGRATION.md §2 (seedable dice: set_seed(); printGameBoard prints the 28
// integers; setGameBoard insists on 24 entries with |count| <= 15).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <array>
#include <cstdint>
#include <iostream>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "bgamd.h"

namespace py = pybind11;

namespace {

enum PLAYERS { PLAYER1 = 0, PLAYER2 = 1 };     // player.hpp:14-18 (unscoped: equals Python ints, as in the reference)

struct Player {                                // player.hpp:6-27
    std::string name;
    PLAYERS num;
    Player(const std::string &n, PLAYERS p) : name(n), num(p) {}
    std::string getName() const { return name; }
    int getNum() const { return (int)num; }
};

const char *const ERR_MESSAGES[8] = {"", "Invalid origin", "Origin out of range", "Destination out of range",
                                     "Cannot move in that direction.", "Move does not match dice.", "Invalid destination.",
                                     "Cannot bear off from jail"};   // game.cpp:585-642, in source order

uint64_t g_seed = std::random_device{}();      // reference: random_device per Game (game.hpp:45)
uint64_t g_next_id = 0;
std::vector<bgamd_env *> g_pool;               // idle one-lane envs: clone() per candidate must not cost an allocation storm

void check(long long rc, const char *what)
{
    if (rc >= 0) return;
    std::string msg = std::string(what) + " failed (" + std::to_string(rc) + "): " + bgamd_error_string((int)rc);
    if (rc == BGAMD_E_HIP) msg += std::string(": ") + bgamd_last_hip_error();
    throw std::runtime_error(msg);
}

class Game;
struct Pieces {                                // Pieces.hpp:8-12 as a live view of its Game
    Game *g;
    int numJailed(int player);
    int numFreed(int player);
};

class Game {
public:
    explicit Game(int player)
    {
        if (!g_pool.empty()) {
            env = g_pool.back();
            g_pool.pop_back();
            check(bgamd_env_reseed(env, g_seed, g_next_id, 1ull << 40, nullptr), "reseed");   // the dice a new env would roll
            check(bgamd_env_reset_stats(env, nullptr), "reset_stats");
        } else
            check(bgamd_env_create(&env, 1, 0, g_seed, g_next_id, 1ull << 40, 32768), "bgamd_env_create");
        ++g_next_id;
        pieces.g = this;
        check(bgamd_game_set_state(env, nullptr, ((player % 2) + 2) % 2), "set_state");   // Game::Game(int): turn = parity (game.cpp:44-53)
    }
    Game(const Game &) = delete;
    Game &operator=(const Game &) = delete;
    ~Game()
    {
        if (!env) return;
        if (g_pool.size() < 64) g_pool.push_back(env);
        else bgamd_env_destroy(env);
    }

    void setPlayers(const Player &p1, const Player &p2) { players[0] = p1; players[1] = p2; has_players = true; }
    Player getPlayer(int num) const
    {
        if (!has_players) throw std::runtime_error("setPlayers has not been called");
        return players[num == 0 ? 0 : 1];                  // by-value copy, as the binding returns
    }
    int getTurn() { return snap()[28]; }
    void setTurn(int turn) { check(bgamd_game_set_state(env, nullptr, turn & 1), "setTurn"); dirty = true; }
    std::vector<int> getGameBoard() { const auto &s = snap(); return std::vector<int>(s.begin(), s.begin() + 24); }
    Pieces &getPieces() { return pieces; }
    int getJailedCount(int player) { return snap()[24 + (player == 0 ? 0 : 1)]; }
    int getBornOffCount(int player) { return snap()[26 + (player == 0 ? 0 : 1)]; }
    void setFreed(int player, int n) { auto s = state28(); s[26 + (player == 0 ? 0 : 1)] = n; put(s); }
    void setGameBoard(const std::vector<int> &b)
    {
        if (b.size() != 24) throw std::invalid_argument("gameboard must have 24 entries");
        auto s = state28();
        for (int i = 0; i < 24; ++i) s[i] = b[i];
        put(s);
    }
    void populateBoard()
    {
        static const int start[24] = {2, 0, 0, 0, 0, -5, 0, -3, 0, 0, 0, 5, -5, 0, 0, 0, 3, 0, 5, 0, 0, 0, 0, -2};   // game.cpp:251
        auto s = state28();
        for (int i = 0; i < 24; ++i) s[i] = start[i];
        put(s);
    }
    void printGameBoard()
    {
        const auto &s = snap();
        std::cout << "board";
        for (int i = 0; i < 24; ++i) std::cout << ' ' << s[i];
        std::cout << " | jail " << s[24] << ' ' << s[25] << " | free " << s[26] << ' ' << s[27] << std::endl;
    }
    void setDice(int d1, int d2) { check(bgamd_game_set_dice(env, d1, d2), "setDice"); dirty = true; }
    std::array<int, 2> rollDice()
    {
        int32_t d[2];
        check(bgamd_game_roll(env, d), "roll_dice");
        dirty = true;
        return {d[0], d[1]};
    }
    std::array<int, 2> getLastDice()
    {
        const auto &s = snap();
        return s[29] ? std::array<int, 2>{s[29], s[30]} : std::array<int, 2>{1, 1};      // last_dice default {1,1} (game.hpp:44)
    }
    std::vector<std::pair<int, int>> legalMoves(int player, int die)
    {
        int8_t pairs[52];
        const int n = bgamd_game_legal_moves(env, player, die, pairs);
        check(n, "legalMoves");
        std::vector<std::pair<int, int>> out;
        for (int i = 0; i < n; ++i) out.emplace_back(pairs[2 * i], pairs[2 * i + 1]);
        return out;
    }
    // (sequences, states int32[N, 28]): evaluateTurnSequencesWrapper, bindings.cpp:19-38
    std::pair<std::vector<std::vector<std::pair<int, int>>>, py::array_t<int32_t>> evaluate(int player, int d1, int d2, bool want_states)
    {
        // one call when the list fits the first guess (p99 of a turn is 1 734 sequences, the median 17), a second one otherwise
        long long cap = 2048, C = 0;
        std::vector<int8_t> seq;
        std::vector<int32_t> len, stv;
        for (int attempt = 0; attempt < 2; ++attempt) {
            seq.resize((size_t)cap * 8); len.resize((size_t)cap);
            if (want_states) stv.resize((size_t)cap * 28);
            C = bgamd_game_enumerate(env, player, d1, d2, want_states ? stv.data() : nullptr, seq.data(), len.data(), cap);
            check(C, "evaluateTurnSequences");
            if (C <= cap) break;
            cap = C;
        }
        py::array_t<int32_t> st({(py::ssize_t)(want_states ? C : 0), (py::ssize_t)28});
        if (want_states && C > 0) std::copy(stv.begin(), stv.begin() + (size_t)C * 28, st.mutable_data());
        std::vector<std::vector<std::pair<int, int>>> seqs((size_t)C);
        for (long long i = 0; i < C; ++i)
            for (int k = 0; k < len[i]; ++k) seqs[i].emplace_back(seq[i * 8 + 2 * k], seq[i * 8 + 2 * k + 1]);
        return {std::move(seqs), std::move(st)};
    }
    std::pair<bool, std::string> tryMove(const Player &p, int dice, int origin, int dest)
    {
        const int code = bgamd_game_try_move(env, p.getNum(), dice, origin, dest);
        check(code, "tryMove");
        if (code == 0) dirty = true;
        return {code == 0, ERR_MESSAGES[code < 8 ? code : 0]};
    }
    std::pair<bool, int> gameOver()
    {
        const int f = snap()[31];
        return (f & 1) ? std::make_pair(true, (f >> 1) & 1) : std::make_pair(false, -1);     // bindings.cpp:11-16
    }
    Game *clone()
    {
        Game *g = new Game(0);
        g->players[0] = players[0]; g->players[1] = players[1]; g->has_players = has_players;
        const auto &s = snap();
        check(bgamd_game_set_state(g->env, s.data(), s[28]), "clone");       // last_dice stays {1,1} (game.cpp:68-77)
        g->dirty = true;
        return g;
    }

private:
    const std::array<int32_t, 32> &snap()
    {
        if (dirty) { check(bgamd_game_snapshot(env, cache.data()), "snapshot"); dirty = false; }
        return cache;
    }
    std::array<int32_t, 28> state28()
    {
        const auto &s = snap();
        std::array<int32_t, 28> o;
        for (int i = 0; i < 28; ++i) o[i] = s[i];
        return o;
    }
    void put(const std::array<int32_t, 28> &s)
    {
        check(bgamd_game_set_state(env, s.data(), -1), "set_state");
        dirty = true;
        uint64_t st[10];
        const int rc = bgamd_env_stats(env, st);                  // |count| > 15 is an error, not a silent wrap
        if (rc < 0) { bgamd_env_reset_stats(env, nullptr); check(rc, "setGameBoard"); }
    }

    bgamd_env *env = nullptr;
    Player players[2] = {Player("", PLAYER1), Player("", PLAYER2)};
    bool has_players = false, dirty = true;
    std::array<int32_t, 32> cache{};
    Pieces pieces{nullptr};
};

int Pieces::numJailed(int player) { return g->getJailedCount(player); }
int Pieces::numFreed(int player) { return g->getBornOffCount(player); }

}  // namespace

PYBIND11_MODULE(backgammon_env, m)
{
    m.doc() = "Backgammon game environment for Reinforcement Learning (MI355X: libbgamd.so behind the reference's module surface)";
    m.def("set_seed", [](uint64_t s) { g_seed = s; g_next_id = 0; }, "Seed the dice of Games created afterwards (the reference cannot be seeded)");
    m.def("source_hash", []() { return std::string(bgamd_source_hash()); });

    py::enum_<PLAYERS>(m, "PlayerType").value("PLAYER1", PLAYER1).value("PLAYER2", PLAYER2);

    py::class_<Player>(m, "Player")
        .def(py::init<const std::string &, PLAYERS>())
        .def("getName", &Player::getName)
        .def("getNum", &Player::getNum);

    py::class_<Pieces>(m, "Pieces").def("numJailed", &Pieces::numJailed).def("numFreed", &Pieces::numFreed);

    py::class_<Game>(m, "Game")
        .def(py::init<int>())
        .def("setPlayers", &Game::setPlayers)
        .def("getPlayers", &Game::getPlayer)
        .def("getTurn", &Game::getTurn)
        .def("setTurn", &Game::setTurn)
        .def("getGameBoard", &Game::getGameBoard)
        .def("getPieces", &Game::getPieces, py::return_value_policy::reference)
        .def("legalMoves", &Game::legalMoves)
        .def("legalTurnSequences", [](Game &g, int player, int d1, int d2) { return g.evaluate(player, d1, d2, false).first; })
        .def("evaluateTurnSequences", [](Game &g, int player, int d1, int d2) { auto r = g.evaluate(player, d1, d2, true); return py::make_tuple(std::move(r.first), std::move(r.second)); },
             "Enumerate all legal turn sequences and their resulting states in one call. Returns (sequences, states[N,28]).")
        .def("tryMove", [](Game &g, const Player &p, int dice, int o, int d) { auto r = g.tryMove(p, dice, o, d); return py::make_tuple(r.first, r.second); })
        .def("is_game_over", [](Game &g) { auto r = g.gameOver(); return py::make_tuple(r.first, r.second); })
        .def("clone", &Game::clone, py::return_value_policy::take_ownership)
        .def("getJailedCount", &Game::getJailedCount)
        .def("setBorneOffPieces", &Game::setFreed)
        .def("getBornOffCount", &Game::getBornOffCount)
        .def("setGameBoard", &Game::setGameBoard)
        .def("setDice", &Game::setDice)
        .def("printGameBoard", &Game::printGameBoard)
        .def("reset", &Game::populateBoard)
        .def("populateBoard", &Game::populateBoard)
        .def("roll_dice", &Game::rollDice, "Roll two dice and return an array [die1, die2]")
        .def("get_last_dice", &Game::getLastDice, "Return the most recently rolled dice as [die1, die2]");

    // the idle one-lane envs must be destroyed while the HIP runtime is still up
    m.add_object("_cleanup", py::capsule([]() { for (bgamd_env *e : g_pool) bgamd_env_destroy(e); g_pool.clear(); }));
}
