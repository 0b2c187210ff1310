// host_selftest — unit tests of the C++ host mirror, restating the reference's own unit tests
// (src/graph.rs:201-227 Handle, src/graph_ops.rs:2051-2132 graph/GFA text, src/layout.rs:262-340,
// src/ygs.rs:247-263 params) plus the fixture parameter table.  Prints "ok <name>" per test;
// exits non-zero on the first failure.  Needs no GPU.
#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>

#include "sgd.hpp"

using namespace gfasort;
#define REQUIRE(c) do { if (!(c)) { std::cerr << "FAIL " << __FILE__ << ":" << __LINE__ << " " #c "\n"; return 1; } } while (0)

static std::string slurp(const std::string &p) { std::ifstream f(p, std::ios::binary); std::ostringstream s; s << f.rdbuf(); return s.str(); }

int main(int argc, char **argv) {
    std::string data = argc > 1 ? argv[1] : "tests/data";
    {   // graph.rs:205-227
        Handle h1 = Handle::forward(42);
        REQUIRE(h1.node_id() == 42 && !h1.is_reverse() && h1.orientation_char() == '+');
        Handle h2 = Handle::reverse(42);
        REQUIRE(h2.node_id() == 42 && h2.is_reverse() && h2.orientation_char() == '-');
        REQUIRE(h2.as_u64() == 85 && Handle::from_u64(85) == h2);
        Handle h3 = Handle::forward(10).flip();
        REQUIRE(h3.node_id() == 10 && h3.is_reverse() && h3.flip() == Handle::forward(10));
        REQUIRE(Handle::reverse(7).to_string() == "7-");
        std::cout << "ok handle\n";
    }
    {   // graph build + GFA text (graph_ops.rs:2051-2132)
        BidirectedGraph g;
        g.add_node(1, "AAAA"); g.add_node(2, "CCCC"); g.add_node(3, "GGGG");
        g.add_edge(Handle::forward(1), Handle::forward(2));
        g.add_edge(Handle::forward(2), Handle::reverse(3));
        g.add_edge(Handle::forward(3), Handle::reverse(2));          // complement of 2+ -> 3-: not added
        REQUIRE(g.node_count() == 3 && g.edges.size() == 2);
        REQUIRE(g.has_edge(Handle::forward(3), Handle::reverse(2)));
        BiPath p; p.name = "test_path"; p.add_step(Handle::forward(1)); p.add_step(Handle::forward(2)); p.add_step(Handle::reverse(3));
        g.paths.push_back(p);
        std::ostringstream out; g.write_gfa(out);
        std::string t = out.str();
        REQUIRE(t.find("H\tVN:Z:1.0\n") == 0);
        REQUIRE(t.find("S\t1\tAAAA\n") != std::string::npos);
        REQUIRE(t.find("L\t1\t+\t2\t+\t0M\n") != std::string::npos);
        REQUIRE(t.find("L\t2\t+\t3\t-\t0M\n") != std::string::npos);
        REQUIRE(t.find("P\ttest_path\t1+,2+,3-\t*\n") != std::string::npos);
        g.add_node(2, "TT");                                          // overwrite keeps node_order
        REQUIRE((g.node_order == std::vector<size_t>{1, 2, 3}));
        // apply_ordering: new id = rank + 1 (graph_ops.rs:1956); node_order untouched
        g.apply_ordering({Handle::forward(3), Handle::forward(1), Handle::forward(2)});
        REQUIRE(g.nodes[1]->sequence == "GGGG" && g.nodes[2]->sequence == "AAAA" && g.nodes[3]->sequence == "TT");
        REQUIRE(g.nodes[1]->rank.value() == 0 && g.paths[0].steps[0] == Handle::forward(2) && g.paths[0].steps[2] == Handle::reverse(1));
        REQUIRE(g.has_edge(Handle::forward(2), Handle::forward(3)) && g.edges.size() == 2);
        REQUIRE((g.node_order == std::vector<size_t>{1, 2, 3}));
        g.apply_ordering({});                                         // no-op
        REQUIRE(g.node_count() == 3);
        std::cout << "ok graph\n";
    }
    {   // layout.rs:262-340
        Layout l(2, 10);
        REQUIRE(l.coords.size() == 40);
        Layout m(2, 5);
        m.set(2, 0, 0, 100); m.set(2, 0, 1, 200); m.set(2, 1, 0, 150); m.set(2, 1, 1, 250);
        REQUIRE(m.get(2, 0, 0) == 100 && m.get(2, 0, 1) == 200 && m.get(2, 1, 0) == 150 && m.get(2, 1, 1) == 250);
        Layout d(2, 2); d.set(1, 0, 0, 3); d.set(1, 0, 1, 4);
        REQUIRE(std::fabs(d.distance(0, 0, 1, 0) - 5.0) < 1e-10);
        Layout v = Layout::from_vectors({{1, 2, 3, 4}, {10, 20, 30, 40}});
        REQUIRE(v.num_nodes == 2 && v.dimensions == 2 && v.get(0, 0, 0) == 1 && v.get(0, 0, 1) == 10 && v.get(0, 1, 0) == 2
                && v.get(0, 1, 1) == 20 && v.get(1, 0, 0) == 3 && v.get(1, 0, 1) == 30);
        Layout t(2, 1); t.set(0, 0, 0, 1.5); t.set(0, 0, 1, 10.0); t.set(0, 1, 0, 0.1 + 0.2); t.set(0, 1, 1, 1e21);
        std::ostringstream out; t.write_tsv(out);
        REQUIRE(out.str() == "idx\tx+\ty+\tx-\ty-\n0\t1.5\t10\t0.30000000000000004\t1000000000000000000000\n");
        REQUIRE(rust_display_f64(1e-7) == "0.0000001" && rust_display_f64(-0.5) == "-0.5");
        std::cout << "ok layout\n";
    }
    {   // ygs.rs:247-263, sgd.rs:214-234,709-729
        YgsParams y;
        REQUIRE(y.path_sgd.iter_max == 100 && y.path_sgd.theta == 0.99 && y.path_sgd.eps == 0.01);
        PathSGDParams d;
        REQUIRE(d.min_term_updates == 100 && d.eta_max == 100.0 && d.space == 100 && d.space_max == 100 && d.seed == 9399220);
        LayoutSGDParams lp;
        REQUIRE(lp.dimensions == 2 && lp.iter_max == 30 && lp.space_max == 1000);
        BidirectedGraph g;
        g.add_node(1, "AAAA"); g.add_node(2, "CCCC"); g.add_node(3, "GGGG");
        BiPath p; p.name = "t"; p.add_step(Handle::forward(1)); p.add_step(Handle::forward(2)); p.add_step(Handle::forward(3));
        g.paths.push_back(p);
        YgsParams f = YgsParams::from_graph(g, 0, 1);
        REQUIRE(f.path_sgd.min_term_updates == 3 && f.path_sgd.eta_max == 9.0 && f.path_sgd.space == 12);
        std::cout << "ok params\n";
    }
    struct Row { const char *name; size_t nodes, steps, paths; uint64_t ymin; double yeta; uint64_t yspace, lmin; uint64_t lspace; };
    Row table[] = { {"simple.gfa", 15, 10, 1, 10, 100.0, 50, 100, 10}, {"lil.gfa", 15, 30, 3, 30, 100.0, 50, 300, 10},
                    {"DRB1-3123.gfa", 4955, 35059, 12, 35059, 9610000.0, 15931, 350590, 3100} };
    for (const Row &r : table) {
        std::string txt = slurp(data + "/" + r.name);
        REQUIRE(!txt.empty());
        BidirectedGraph g = parse_gfa(txt);
        FlatGraph f = g.flatten();
        REQUIRE(g.node_count() == r.nodes && f.step_node.size() == r.steps && g.paths.size() == r.paths);
        YgsParams y = YgsParams::from_graph(g, 0, 1);
        REQUIRE(y.path_sgd.min_term_updates == r.ymin && y.path_sgd.eta_max == r.yeta && y.path_sgd.space == r.yspace);
        LayoutSGDParams lp = LayoutSGDParams::from_graph(g, 2, 1);
        REQUIRE(lp.min_term_updates == r.lmin && lp.eta_max == r.yeta && lp.space == r.lspace);
        // write -> reload keeps counts (integration_tests.rs:175-206)
        std::ostringstream out; g.write_gfa(out);
        BidirectedGraph g2 = parse_gfa(out.str());
        REQUIRE(g2.node_count() == g.node_count() && g2.edges.size() == g.edges.size() && g2.paths.size() == g.paths.size());
        // the C ABI's host tables are callable without a GPU
        gfs_sgd_params cp = y.path_sgd.to_c();
        std::vector<double> etas(cp.iter_max + 1);
        REQUIRE(gfs_sgd_schedule(&cp, etas.data()) == 0 && etas[0] == 1.0 / (1.0 / r.yeta));
        std::cout << "ok fixture " << r.name << "\n";
    }
    {   // empty graph: SGD entry points return the empty result before touching the device
        BidirectedGraph g;
        REQUIRE(path_linear_sgd(g, PathSGDParams()).empty());
        REQUIRE(path_sgd_sort(g, PathSGDParams()).empty());
        Layout l = path_linear_sgd_layout(g, LayoutSGDParams());
        REQUIRE(l.num_nodes == 0 && l.dimensions == 2);
        std::cout << "ok empty\n";
    }
    {   // the reader's path-step tokenizer: trims, skips empty tokens, cuts long lists into pieces (any thread count)
        std::string small = "S\t1\tA\nS\t2\tC\nS\t3\tG\nP\tp\t1+,, 2- ,3+ ,\t*\nP\tempty\t\t*\nP\tq\t3x,+2+\t*\n";
        for (size_t nt : {1, 3}) {
            set_io_threads(nt);
            BidirectedGraph g = parse_gfa(small);
            REQUIRE(g.paths.size() == 3 && g.paths[1].steps.empty());
            REQUIRE((g.paths[0].steps == std::vector<Handle>{Handle::forward(1), Handle::reverse(2), Handle::forward(3)}));
            // any last character other than '+' reads as reverse; a leading '+' is part of the number (usize::from_str)
            REQUIRE((g.paths[2].steps == std::vector<Handle>{Handle::reverse(3), Handle::forward(2)}));
            std::ostringstream out; g.write_gfa(out);
            REQUIRE(out.str().find("P\tp\t1+,2-,3+\t*\nP\tempty\t\t*\nP\tq\t3-,2+\t*\n") != std::string::npos);
            // 700k steps (several pieces), an empty token in every 1000th place so that pieces have to be closed up
            std::string big = "S\t1\tA\nP\tlong\t";
            std::vector<Handle> want;
            for (size_t k = 0; k < 700000; ++k) {
                size_t id = 1 + (k * 2654435761u) % 999983;
                bool rev = (k % 7) == 3;
                big += std::to_string(id); big += rev ? '-' : '+'; big += ',';
                if (k % 1000 == 999) big += " ,";
                want.push_back(Handle::make(id, rev));
            }
            big += "\t*\n";
            BidirectedGraph gb = parse_gfa(big);
            REQUIRE(gb.paths.size() == 1 && gb.paths[0].steps == want);
            FlatGraph f = gb.flatten();
            REQUIRE(f.step_node.size() == want.size() && f.step_is_rev[3] == 1 && f.step_node[0] == 0 && f.step_node[1] == GFS_NO_NODE);
            std::ostringstream ob; gb.write_gfa(ob);
            BidirectedGraph gb2 = parse_gfa(ob.str());
            REQUIRE(gb2.paths[0].steps == want);
            // a bad token in a late piece is reported with the reference's text
            std::string bad = big; bad.replace(bad.size() - 20, 1, "x");
            bool threw = false;
            try { (void)parse_gfa(bad); } catch (const std::exception &e) {
                threw = std::string(e.what()) == "Failed to parse path node ID: invalid digit found in string";
            }
            REQUIRE(threw);
            threw = false;
            try { (void)parse_gfa("P\tp\t99999999999999999999+\t*\n"); } catch (const std::exception &e) {
                threw = std::string(e.what()) == "Failed to parse path node ID: number too large to fit in target type";
            }
            REQUIRE(threw);
            REQUIRE(parse_gfa("P\tp\t18446744073709551615+\t*\n").paths[0].steps.size() == 1);   // u64::MAX parses
        }
        set_io_threads(0);
        std::cout << "ok tokenizer\n";
    }
    {   // EdgeSet: set semantics under growth, complement rule of add_edge, ordered L lines
        BidirectedGraph g;
        std::vector<BiEdge> list;
        for (size_t k = 1; k <= 5000; ++k) {
            list.push_back(BiEdge{Handle::forward(k), Handle::forward(k + 1)});
            list.push_back(BiEdge{Handle::reverse(k + 1), Handle::reverse(k)});      // complement: not added
            list.push_back(BiEdge{Handle::forward(k), Handle::forward(k + 1)});      // duplicate
        }
        g.add_edges(list);
        REQUIRE(g.edges.size() == 5000);
        size_t seen = 0;
        for (const BiEdge &e : g.edges) { REQUIRE(e.to.node_id() == e.from.node_id() + 1); ++seen; }
        REQUIRE(seen == 5000 && g.has_edge(Handle::reverse(78), Handle::reverse(77)) && !g.has_edge(Handle::forward(78), Handle::forward(77)));
        std::ostringstream out; g.write_gfa(out);
        REQUIRE(out.str().find("L\t1\t+\t2\t+\t0M\nL\t2\t+\t3\t+\t0M\n") != std::string::npos);
        std::cout << "ok edges\n";
    }
    std::cout << "ALL OK\n";
    return 0;
}
