// Math-expression evaluator for input decks: what the reference gets from amrex::Parser
// (un-vendored AMReX, Src/Base/Parser) through utils::parser::makeParser / queryWithParser
// (Source/Utils/Parser/ParserUtils.cpp).  Written from the documented grammar
// (Docs/source/usage/parameters.rst, "Math parser and user-defined constants"):
//   numbers, names, + - * / ^ **, unary + -, < > <= >= == !=, and or, parentheses, and the functions
//   sqrt exp log log10 sin cos tan asin acos atan sinh cosh tanh abs fabs floor ceil erf
//   pow atan2 min max fmod heaviside if(a,b,c)
// with the precedence of a C expression and '^' / '**' binding tighter than unary minus (-2^2 = -4).
// An expression is compiled once into a postfix program and evaluated with a small value stack.
#ifndef WXA_HOST_PARSER_HPP_
#define WXA_HOST_PARSER_HPP_

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace wxa::host {

class Parser {
public:
    Parser() = default;
    // `vars`: names bound at evaluation time, in the order of eval()'s array; `constants`: names folded now
    Parser(const std::string& expr, const std::vector<std::string>& vars, const std::map<std::string, double>& constants)
        : m_src(expr), m_vars(vars), m_consts(&constants) {
        m_pos = 0;
        parse_or();
        skip_ws();
        if (m_pos != m_src.size()) fail("unexpected '" + m_src.substr(m_pos, 8) + "'");
        m_consts = nullptr;
        int depth = 0, deepest = 0;   // eval() uses a fixed value stack
        for (const Op& op : m_prog) {
            depth += (op.code == NUM || op.code == VAR) ? 1 : (op.code == NEG || op.code == F1) ? 0 : (op.code == IF) ? -2 : -1;
            deepest = depth > deepest ? depth : deepest;
        }
        if (deepest > 64) fail("expression too deep");
    }

    bool empty() const { return m_prog.empty(); }

    double eval(const double* vals) const {
        double st[64];
        int sp = 0;
        for (const Op& op : m_prog) {
            switch (op.code) {
                case NUM: st[sp++] = op.value; break;
                case VAR: st[sp++] = vals[op.arg]; break;
                case NEG: st[sp - 1] = -st[sp - 1]; break;
                case ADD: st[sp - 2] = st[sp - 2] + st[sp - 1]; --sp; break;
                case SUB: st[sp - 2] = st[sp - 2] - st[sp - 1]; --sp; break;
                case MUL: st[sp - 2] = st[sp - 2] * st[sp - 1]; --sp; break;
                case DIV: st[sp - 2] = st[sp - 2] / st[sp - 1]; --sp; break;
                case POW: st[sp - 2] = power(st[sp - 2], st[sp - 1]); --sp; break;
                case LT: st[sp - 2] = st[sp - 2] < st[sp - 1] ? 1.0 : 0.0; --sp; break;
                case GT: st[sp - 2] = st[sp - 2] > st[sp - 1] ? 1.0 : 0.0; --sp; break;
                case LE: st[sp - 2] = st[sp - 2] <= st[sp - 1] ? 1.0 : 0.0; --sp; break;
                case GE: st[sp - 2] = st[sp - 2] >= st[sp - 1] ? 1.0 : 0.0; --sp; break;
                case EQ: st[sp - 2] = st[sp - 2] == st[sp - 1] ? 1.0 : 0.0; --sp; break;
                case NE: st[sp - 2] = st[sp - 2] != st[sp - 1] ? 1.0 : 0.0; --sp; break;
                case AND: st[sp - 2] = (st[sp - 2] != 0.0 && st[sp - 1] != 0.0) ? 1.0 : 0.0; --sp; break;
                case OR: st[sp - 2] = (st[sp - 2] != 0.0 || st[sp - 1] != 0.0) ? 1.0 : 0.0; --sp; break;
                case F1: st[sp - 1] = call1(op.arg, st[sp - 1]); break;
                case F2: st[sp - 2] = call2(op.arg, st[sp - 2], st[sp - 1]); --sp; break;
                case IF: st[sp - 3] = st[sp - 3] != 0.0 ? st[sp - 2] : st[sp - 1]; sp -= 2; break;
            }
        }
        return st[0];
    }
    double eval() const { return eval(nullptr); }

private:
    enum Code { NUM, VAR, NEG, ADD, SUB, MUL, DIV, POW, LT, GT, LE, GE, EQ, NE, AND, OR, F1, F2, IF };
    struct Op { Code code; int arg; double value; };
    enum F1Id { SQRT, EXP, LOG, LOG10, SIN, COS, TAN, ASIN, ACOS, ATAN, SINH, COSH, TANH, ABS, FLOOR, CEIL, ERF };
    enum F2Id { FPOW, ATAN2, FMIN, FMAX, FMOD, HEAVISIDE };

    // integer exponents by repeated multiplication, like the reference's parser does for constant integer powers
    static double power(double a, double b) {
        if (b == std::floor(b) && std::fabs(b) <= 16.0) {
            int n = (int)std::fabs(b);
            double r = 1.0, x = a;
            while (n) { if (n & 1) r *= x; x *= x; n >>= 1; }
            return b < 0 ? 1.0 / r : r;
        }
        return std::pow(a, b);
    }
    static double call1(int id, double a) {
        switch (id) {
            case SQRT: return std::sqrt(a);   case EXP: return std::exp(a);     case LOG: return std::log(a);
            case LOG10: return std::log10(a); case SIN: return std::sin(a);     case COS: return std::cos(a);
            case TAN: return std::tan(a);     case ASIN: return std::asin(a);   case ACOS: return std::acos(a);
            case ATAN: return std::atan(a);   case SINH: return std::sinh(a);   case COSH: return std::cosh(a);
            case TANH: return std::tanh(a);   case ABS: return std::fabs(a);    case FLOOR: return std::floor(a);
            case CEIL: return std::ceil(a);   case ERF: return std::erf(a);
        }
        return 0.0;
    }
    static double call2(int id, double a, double b) {
        switch (id) {
            case FPOW: return power(a, b);   case ATAN2: return std::atan2(a, b);
            case FMIN: return a < b ? a : b; case FMAX: return a > b ? a : b;
            case FMOD: return std::fmod(a, b);
            case HEAVISIDE: return a < 0.0 ? 0.0 : (a > 0.0 ? 1.0 : b);
        }
        return 0.0;
    }

    [[noreturn]] void fail(const std::string& what) const {
        throw std::runtime_error("parser: " + what + " in \"" + m_src + "\"");
    }
    void skip_ws() { while (m_pos < m_src.size() && std::isspace((unsigned char)m_src[m_pos])) ++m_pos; }
    bool eat(const char* tok) {
        skip_ws();
        const size_t n = std::strlen(tok);
        if (m_src.compare(m_pos, n, tok) != 0) return false;
        m_pos += n;
        return true;
    }
    bool eat_word(const char* w) {   // a whole identifier
        skip_ws();
        const size_t n = std::strlen(w);
        if (m_src.compare(m_pos, n, w) != 0) return false;
        const size_t e = m_pos + n;
        if (e < m_src.size() && (std::isalnum((unsigned char)m_src[e]) || m_src[e] == '_')) return false;
        m_pos = e;
        return true;
    }
    void emit(Code c, int arg = 0, double v = 0.0) { m_prog.push_back(Op{c, arg, v}); }

    void parse_or() {
        parse_and();
        while (eat_word("or")) { parse_and(); emit(OR); }
    }
    void parse_and() {
        parse_eq();
        while (eat_word("and")) { parse_eq(); emit(AND); }
    }
    void parse_eq() {
        parse_rel();
        for (;;) {
            if (eat("==")) { parse_rel(); emit(EQ); }
            else if (eat("!=")) { parse_rel(); emit(NE); }
            else break;
        }
    }
    void parse_rel() {
        parse_add();
        for (;;) {
            if (eat("<=")) { parse_add(); emit(LE); }
            else if (eat(">=")) { parse_add(); emit(GE); }
            else if (eat("<")) { parse_add(); emit(LT); }
            else if (eat(">")) { parse_add(); emit(GT); }
            else break;
        }
    }
    void parse_add() {
        parse_mul();
        for (;;) {
            if (eat("+")) { parse_mul(); emit(ADD); }
            else if (eat("-")) { parse_mul(); emit(SUB); }
            else break;
        }
    }
    void parse_mul() {
        parse_unary();
        for (;;) {
            skip_ws();
            if (m_src.compare(m_pos, 2, "**") == 0) break;   // belongs to parse_pow
            if (eat("*")) { parse_unary(); emit(MUL); }
            else if (eat("/")) { parse_unary(); emit(DIV); }
            else break;
        }
    }
    void parse_unary() {
        if (eat("-")) { parse_unary(); emit(NEG); }
        else if (eat("+")) { parse_unary(); }
        else parse_pow();
    }
    void parse_pow() {   // right associative, exponent may carry its own sign
        parse_primary();
        if (eat("**") || eat("^")) { parse_unary(); emit(POW); }
    }
    void parse_primary() {
        skip_ws();
        if (m_pos >= m_src.size()) fail("unexpected end");
        const char c = m_src[m_pos];
        if (c == '(') {
            ++m_pos;
            parse_or();
            if (!eat(")")) fail("missing ')'");
            return;
        }
        if (std::isdigit((unsigned char)c) || c == '.') {
            const char* b = m_src.c_str() + m_pos;
            char* e = nullptr;
            const double v = std::strtod(b, &e);
            if (e == b) fail("bad number");
            m_pos += (size_t)(e - b);
            emit(NUM, 0, v);
            return;
        }
        if (std::isalpha((unsigned char)c) || c == '_') {
            size_t e = m_pos;
            while (e < m_src.size() && (std::isalnum((unsigned char)m_src[e]) || m_src[e] == '_')) ++e;
            const std::string name = m_src.substr(m_pos, e - m_pos);
            m_pos = e;
            skip_ws();
            if (m_pos < m_src.size() && m_src[m_pos] == '(') { ++m_pos; parse_call(name); return; }
            for (size_t i = 0; i < m_vars.size(); ++i)
                if (m_vars[i] == name) { emit(VAR, (int)i); return; }
            const auto it = m_consts->find(name);
            if (it == m_consts->end()) fail("unknown name '" + name + "'");
            emit(NUM, 0, it->second);
            return;
        }
        fail(std::string("unexpected '") + c + "'");
    }
    int parse_args() {
        int n = 0;
        skip_ws();
        if (eat(")")) return 0;
        for (;;) {
            parse_or();
            ++n;
            if (eat(",")) continue;
            if (eat(")")) return n;
            fail("missing ')' in a function call");
        }
    }
    void parse_call(const std::string& name) {
        static const std::map<std::string, int> f1 = {
            {"sqrt", SQRT}, {"exp", EXP}, {"log", LOG}, {"log10", LOG10}, {"sin", SIN}, {"cos", COS}, {"tan", TAN},
            {"asin", ASIN}, {"acos", ACOS}, {"atan", ATAN}, {"sinh", SINH}, {"cosh", COSH}, {"tanh", TANH},
            {"abs", ABS}, {"fabs", ABS}, {"floor", FLOOR}, {"ceil", CEIL}, {"erf", ERF}};
        static const std::map<std::string, int> f2 = {
            {"pow", FPOW}, {"atan2", ATAN2}, {"min", FMIN}, {"max", FMAX}, {"fmod", FMOD}, {"heaviside", HEAVISIDE}};
        const int n = parse_args();
        const auto i1 = f1.find(name);
        if (i1 != f1.end()) { if (n != 1) fail(name + " takes 1 argument"); emit(F1, i1->second); return; }
        const auto i2 = f2.find(name);
        if (i2 != f2.end()) { if (n != 2) fail(name + " takes 2 arguments"); emit(F2, i2->second); return; }
        if (name == "if") { if (n != 3) fail("if takes 3 arguments"); emit(IF); return; }
        fail("unknown function '" + name + "'");
    }

    std::string m_src;
    std::vector<std::string> m_vars;
    const std::map<std::string, double>* m_consts = nullptr;
    size_t m_pos = 0;
    std::vector<Op> m_prog;
};

}  // namespace wxa::host
#endif
